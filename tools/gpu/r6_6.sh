cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "ends_with_its_cross_attention" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 )
timeout 600 python3 tools/xattn_bench.py 20 > gpurun_out/r6/xattn_bench.txt 2>&1; head -22 gpurun_out/r6/xattn_bench.txt
