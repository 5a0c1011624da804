R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4n; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -x -q -m gpu -s -k "reduce_pass or that_ends_with or layernorms or groupnorm" 2>&1 | grep -v "^$" | tail -12
timeout 600 python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_r3.so mlimgsynth_amd/lib/libmlimgsynth_amd.so 1 2>&1 | tail -6
