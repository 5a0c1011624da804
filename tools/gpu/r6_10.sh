cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 600 python3 tools/attn_sp_bench.py 20 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > gpurun_out/r6/attn_sp_bench.txt; cat gpurun_out/r6/attn_sp_bench.txt
