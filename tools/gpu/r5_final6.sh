cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh sdxl 4 r5 2>&1 | tail -1 | cut -c1-200
bash tools/profile_round.sh sd15 1 r5 2>&1 | tail -1 | cut -c1-200
mkdir -p gpurun_out/r5; timeout 900 python bench.py > gpurun_out/r5/bench_default_c.json 2> gpurun_out/r5/bench_default_c.err; tail -c 400 gpurun_out/r5/bench_default_c.json
