R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "stream_k or groupnorm" 2>&1 | tail -8
timeout 900 python3 tools/tune_inplan.py $O/tune_sd15.inc unet:sd1:64:2 > $O/tune_sd15.log 2>&1; tail -40 $O/tune_sd15.log
