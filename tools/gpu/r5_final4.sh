cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/r5/bench_default_b.json 2> gpurun_out/r5/bench_default_b.err; tail -c 1500 gpurun_out/r5/bench_default_b.json; tail -3 gpurun_out/r5/bench_default_b.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
