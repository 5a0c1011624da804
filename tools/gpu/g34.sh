R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 900 python3 tools/soak_stream.py 300 8 2>&1 | grep -v amdgpu | tee $O/r4_soak_stream.txt
