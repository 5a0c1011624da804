R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4w; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python3 -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 1500 python3 bench.py --steps 5 --warmup 2 > $O/r4_bench_default_line.json 2> $O/bench_err.log; tail -c 150 $O/r4_bench_default_line.json
