cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | grep -v "$F" | tail -6 ) > gpurun_out/r6/t14_attn_tests.txt; tail -3 gpurun_out/r6/t14_attn_tests.txt
( timeout 1500 python -m pytest tests/test_golden_gpu.py tests/test_unet_gpu.py tests/test_determinism_gpu.py -x -q -s -k "sdxl or tinyxl or determinism or repeatable" 2>&1 | grep -E "rel-L2|unet_|passed|failed|Error|error|^E " | tail -30 ) > gpurun_out/r6/t14_unet_tests.txt; tail -30 gpurun_out/r6/t14_unet_tests.txt
LIB=mlimgsynth_amd/lib/libmlimgsynth_amd.so
AB_ONLY=sdxl_b4 AB_ENV_A=MLSD_ATTN_SP=0 timeout 1200 python3 tools/ab_eval.py $LIB $LIB 3 > gpurun_out/r6/ab_attn_sp.txt 2>&1; tail -4 gpurun_out/r6/ab_attn_sp.txt
