R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 300 python3 tools/h2d_probe.py 1024 2>&1 | grep -v amdgpu | tee $O/r4_h2d_probe.txt
