R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "skinny" 2>&1 | tail -12
timeout 600 python3 tools/gemm_skinny.py 2>&1 | tee $O/gemm_skinny.txt | tail -70
