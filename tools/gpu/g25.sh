R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 1500 python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_head.so mlimgsynth_amd/lib/libmlimgsynth_amd.so 2 2>&1 | grep -v amdgpu | tee $O/ab_head_vs_sk2.txt
