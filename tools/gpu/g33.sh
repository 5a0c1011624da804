R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "stream or split" 2>&1 | tail -8
