R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4l; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 900 python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_r3.so mlimgsynth_amd/lib/libmlimgsynth_amd.so 2 2>&1 | tee $O/r4_ab_r3_vs_r4.txt
