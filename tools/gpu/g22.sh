R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R; export PYTHONPATH=$R
for s in "8192 320 5760" "8192 320 2880" "2048 640 5760" "512 1280 11520" "8192 1280 1280"; do timeout 120 python3 tools/gemm_trace_sk.py $s 28 2>&1 | grep -v amdgpu; done | tee $O/r4_gemm_trace_sk.txt
