R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
timeout 1200 python3 -m pytest tests/test_pipeline_gpu.py tests/test_sampler_gpu.py tests/test_determinism_gpu.py tests/test_api_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
