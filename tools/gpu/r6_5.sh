cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for f in 8 0; do timeout 300 python3 tools/two_stream_eval.py sd1 64 2 30 $f 2>&1 | tail -2; done > gpurun_out/r6/two_stream_sd15.txt
cat gpurun_out/r6/two_stream_sd15.txt
timeout 300 python3 tools/two_stream_eval.py sd1 64 4 30 8 2>&1 | tail -2
