R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 1500 python3 bench.py --steps 5 --warmup 2 > $O/r4_bench_default_line.json 2> $O/bench_err.log; tail -c 200 $O/r4_bench_default_line.json
