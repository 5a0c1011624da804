cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 1800 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm or two_tiles or ends_with" 2>&1 | grep -v "$F" | tail -8 ) > gpurun_out/r6/t17_ln_tests.txt; tail -4 gpurun_out/r6/t17_ln_tests.txt
for i in 1 2 3 4 5; do ( timeout 900 python3 tools/soak_r5.py 3000 10 2>&1 | grep -v "$F" ) > gpurun_out/r6/soak_e$i.txt; grep "MISMATCH\|COUNTERS\|soak" gpurun_out/r6/soak_e$i.txt | head -6 | cut -c1-200; done
( timeout 900 python3 tools/soak_r5.py 3000 10 pp 2>&1 | grep -v "$F" ) > gpurun_out/r6/soak_e_pp.txt; grep "MISMATCH\|COUNTERS\|soak" gpurun_out/r6/soak_e_pp.txt | head -6 | cut -c1-200
( timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_determinism_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | grep -v "$F" | tail -5 ) > gpurun_out/r6/t17_unet_tests.txt; tail -3 gpurun_out/r6/t17_unet_tests.txt
timeout 300 python3 tools/ln_fold_bench.py 2>&1 | grep -v "$F" | tail -3
