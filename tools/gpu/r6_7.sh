# round 6, call 7: LayerNorm endings on self-tagged records (no ticket / arrival / departure counters)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( timeout 1800 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm or two_tiles or ends_with" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 ) > gpurun_out/r6/t7_ln_tests.txt
tail -6 gpurun_out/r6/t7_ln_tests.txt
timeout 300 python3 tools/ln_fold_bench.py > gpurun_out/r6/ln_fold_bench.txt 2>&1; cat gpurun_out/r6/ln_fold_bench.txt | tail -3
( timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_determinism_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 ) > gpurun_out/r6/t7_unet_tests.txt
tail -4 gpurun_out/r6/t7_unet_tests.txt
LIB=mlimgsynth_amd/lib/libmlimgsynth_amd.so
timeout 1500 python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_r4.so $LIB 2 > gpurun_out/r6/ab_r4_vs_r6.txt 2>&1; tail -9 gpurun_out/r6/ab_r4_vs_r6.txt
