R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 300 python3 tools/wstream_trace.py sdxl 128 8 512 > $O/r4_wstream_trace.txt 2>&1; echo rc=$?; tail -25 $O/r4_wstream_trace.txt
