R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 1500 python3 -m pytest tests -q -m gpu --durations=15 -p no:cacheprovider > $O/pytest_full.log 2>&1; tail -40 $O/pytest_full.log
