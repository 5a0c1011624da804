cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/profiles_r5
python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_r4.so mlimgsynth_amd/lib/libmlimgsynth_amd.so 3 > gpurun_out/profiles_r5/r5_ab_r4_vs_r5.txt 2>&1
tail -8 gpurun_out/profiles_r5/r5_ab_r4_vs_r5.txt
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gpu_tests.log 2>&1; tail -3 gpurun_out/r5/gpu_tests.log
