R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 600 python3 tools/soak_r4.py 2>&1 | grep -v amdgpu | tail -12 | tee $O/soak.txt
timeout 1500 python3 bench.py --steps 5 --warmup 2 > $O/r4_bench_default_line.json 2> $O/bench_err.log; tail -c 400 $O/r4_bench_default_line.json
