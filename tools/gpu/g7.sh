R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 1200 python3 bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench_err.log; tail -c 6000 $O/bench.json; tail -5 $O/bench_err.log
