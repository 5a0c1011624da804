R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 600 python3 tools/two_stream_eval.py sdxl 128 8 6 2>&1 | grep -v amdgpu | tee $O/r4_two_stream_sdxl.txt
timeout 600 python3 tools/two_stream_eval.py sdxl 128 16 4 2>&1 | grep -v amdgpu | tee -a $O/r4_two_stream_sdxl.txt
