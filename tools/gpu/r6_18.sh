cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "geglu or gelu or act" 2>&1 | grep -v "$F" | tail -4 ) > gpurun_out/r6/t18_gelu_tests.txt; tail -2 gpurun_out/r6/t18_gelu_tests.txt
( timeout 1500 python -m pytest tests/test_golden_gpu.py tests/test_unet_gpu.py -x -q -s -k "sdxl or sd1 or tiny or clip" 2>&1 | grep -E "rel-L2|unet_|clip_|passed|failed|Error|^E " | tail -40 ) > gpurun_out/r6/t18_unet_tests.txt; tail -40 gpurun_out/r6/t18_unet_tests.txt
LIB=mlimgsynth_amd/lib/libmlimgsynth_amd.so
AB_ONLY=sdxl_b4 timeout 1200 python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_r6b.so $LIB 3 > gpurun_out/r6/ab_gelu_folded.txt 2>&1; tail -3 gpurun_out/r6/ab_gelu_folded.txt
timeout 300 python3 tools/shape_table.py sdxl 128 8 unet 2>/dev/null | grep "8192x10240x1280\|32768x5120x640"
