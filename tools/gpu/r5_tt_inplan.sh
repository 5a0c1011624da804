# in-plan A/B of the two-tiles-per-CU GEMM (MLSD_TT): SDXL b4 evaluation, alternating processes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
LIB=mlimgsynth_amd/lib/libmlimgsynth_amd.so
(AB_ONLY=sdxl_b4 AB_ENV_B=MLSD_TT=2 python3 tools/ab_eval.py $LIB $LIB 2; AB_ONLY=sdxl_b4 AB_ENV_B=MLSD_TT=1 python3 tools/ab_eval.py $LIB $LIB 2) > gpurun_out/r5/tt_inplan.txt 2>&1
tail -20 gpurun_out/r5/tt_inplan.txt
