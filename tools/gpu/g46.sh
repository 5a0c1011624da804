R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm or column_statistics" 2>&1 | tail -3
timeout 1500 python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_head.so mlimgsynth_amd/lib/libmlimgsynth_amd.so 2 2>&1 | grep -v amdgpu | tail -7 | tee $O/gn_stats_ab5.txt
