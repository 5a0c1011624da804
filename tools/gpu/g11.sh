R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "skinny" 2>&1 | tail -3
timeout 1500 python3 tools/tune_inplan.py $O/tune_r4.inc unet:sd1:64:2 unet:sdxl:128:8 unet:sdxl:128:16 vae:sdxl:128:4 vae:sd1:64:1 vae:sdxl:128:8 unet:sd1:64:4 unet:sd1:64:8 > $O/tune.log 2>&1; tail -60 $O/tune.log
