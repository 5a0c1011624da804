# A/B of the LayerNorm producers moved to the 128x160 kernel (MLSD_TT_LN) + the untuned-size test + abort test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
LIB=mlimgsynth_amd/lib/libmlimgsynth_amd.so
AB_ENV_A=MLSD_TT_LN=0 python3 tools/ab_eval.py $LIB $LIB 2 > gpurun_out/r5/tt_ln_inplan.txt 2>&1
tail -9 gpurun_out/r5/tt_ln_inplan.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q -s -k "nobody_tuned or layernorms" 2>&1 | grep -E "sdxl b1|LayerNorms|passed|failed|Error|assert" | tail -12
timeout 600 python -m pytest tests/test_sampler_gpu.py -x -q -k "callback" 2>&1 | tail -3
