cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 600 python3 tools/coissue_probe.py 2>&1 | grep -v "$F" > gpurun_out/r6/coissue_probe.txt; cat gpurun_out/r6/coissue_probe.txt
for L in gpurun_lib/libattn_pk1.so mlimgsynth_amd/lib/libmlimgsynth_amd.so; do echo "== $L"; MLSD_LIB_PATH=$L timeout 600 python3 tools/attn_sp_bench.py 20 2>&1 | grep -v "$F"; done > gpurun_out/r6/attn_pk_ab.txt 2>&1
cat gpurun_out/r6/attn_pk_ab.txt
