R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; mkdir -p $O
cd $R; export PYTHONPATH=$R
for f in 0 8; do timeout 200 python3 tools/two_stream_eval.py sd1 64 2 20 $f; done 2>&1 | tee $O/two_stream.txt
MLSD_GN_SINGLE=0 timeout 200 python3 tools/two_stream_eval.py sd1 64 2 20 8 2>&1 | tee -a $O/two_stream.txt
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm or gemm_split or conv2d" 2>&1 | tail -5
timeout 300 python3 tools/shape_table.py sd1 64 2 unet 3 > $O/sd15_shape_table.txt 2> $O/err1.log; head -30 $O/sd15_shape_table.txt
