cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/profiles_r5
timeout 900 python3 tools/soak_r5.py 2000 100 > gpurun_out/profiles_r5/r5_soak.txt 2>&1; tail -8 gpurun_out/profiles_r5/r5_soak.txt
timeout 300 python3 tools/ln_fold_bench.py 2>&1 | tail -3
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm or two_tiles" 2>&1 | tail -3
AB_ONLY=sdxl_b4 python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_r4.so mlimgsynth_amd/lib/libmlimgsynth_amd.so 3 2>&1 | tail -3
