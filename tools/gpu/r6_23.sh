cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 1500 python -m pytest tests/test_golden_gpu.py tests/test_unet_gpu.py tests/test_determinism_gpu.py tests/test_pipeline_gpu.py -x -q -s -k "sd1 or sd15 or tiny or determinism" 2>&1 | grep -E "rel-L2|unet_sd1|passed|failed|Error|^E " | tail -14 ) > gpurun_out/r6/t23_sd1_tests.txt; tail -14 gpurun_out/r6/t23_sd1_tests.txt
LIB=mlimgsynth_amd/lib/libmlimgsynth_amd.so
AB_ONLY=sd15_b1,sd15_b2 AB_ENV_A=MLSD_ATTN_SP=0 timeout 1200 python3 tools/ab_eval.py $LIB $LIB 3 > gpurun_out/r6/ab_attn_sp_sd15.txt 2>&1; tail -4 gpurun_out/r6/ab_attn_sp_sd15.txt
