R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4t; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python3 -m pytest tests -q -m gpu --durations=8 -p no:cacheprovider > $O/pytest_full.log 2>&1; tail -14 $O/pytest_full.log
