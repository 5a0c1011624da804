R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 600 python3 tools/gemm_panel_sweep.py 20 2>&1 | grep -v amdgpu | tee $O/r4_gemm_panel_sweep.txt
