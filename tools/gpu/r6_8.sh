cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
LIB=mlimgsynth_amd/lib/libmlimgsynth_amd.so
AB_ONLY=sdxl_b4 timeout 1200 python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_r6a.so $LIB 3 > gpurun_out/r6/ab_ln_tags.txt 2>&1; tail -3 gpurun_out/r6/ab_ln_tags.txt
timeout 1800 python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_r5.so $LIB 3 > gpurun_out/r6/ab_r5_vs_r6.txt 2>&1; tail -8 gpurun_out/r6/ab_r5_vs_r6.txt
