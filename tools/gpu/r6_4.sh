# round 6, call 4: cross attention at the end of its q projection -- parity, in-plan A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "ends_with_its_cross_attention" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -25 ) > gpurun_out/r6/t4_xattn_tests.txt
tail -12 gpurun_out/r6/t4_xattn_tests.txt
( timeout 1200 python -m pytest tests/test_golden_gpu.py tests/test_unet_gpu.py -x -q -s -k "sdxl or tinyxl" 2>&1 | grep -E "rel-L2|unet_|passed|failed|Error|error" | tail -30 ) > gpurun_out/r6/t4_unet_tests.txt
tail -30 gpurun_out/r6/t4_unet_tests.txt
LIB=mlimgsynth_amd/lib/libmlimgsynth_amd.so
AB_ONLY=sdxl_b4 AB_ENV_A=MLSD_XATTN=0 timeout 1200 python3 tools/ab_eval.py $LIB $LIB 3 > gpurun_out/r6/ab_xattn.txt 2>&1; tail -8 gpurun_out/r6/ab_xattn.txt
timeout 300 python3 tools/shape_table.py sdxl 128 8 unet > gpurun_out/r6/unet_shape_xattn.txt 2>&1; head -3 gpurun_out/r6/unet_shape_xattn.txt; grep "attention" gpurun_out/r6/unet_shape_xattn.txt
MLSD_XATTN=0 timeout 300 python3 tools/shape_table.py sdxl 128 8 unet > gpurun_out/r6/unet_shape_noxattn.txt 2>&1; head -3 gpurun_out/r6/unet_shape_noxattn.txt; grep "attention\|8192x1280x1280\|32768x640x640" gpurun_out/r6/unet_shape_noxattn.txt
