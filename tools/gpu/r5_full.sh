cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/r5/bench_default_a.json 2> gpurun_out/r5/bench_default_a.err; tail -c 6000 gpurun_out/r5/bench_default_a.json; tail -5 gpurun_out/r5/bench_default_a.err
