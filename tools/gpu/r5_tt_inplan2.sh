cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
LIB=mlimgsynth_amd/lib/libmlimgsynth_amd.so
AB_ENV_B=MLSD_TT=3 python3 tools/ab_eval.py $LIB $LIB 2 > gpurun_out/r5/tt_inplan_mode3.txt 2>&1
tail -9 gpurun_out/r5/tt_inplan_mode3.txt
