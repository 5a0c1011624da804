# round 6, call 1: new GPU tests of the bench plumbing + the r6 PMC / kernel-stat profiles + baselines for the round's kernel work
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( timeout 1500 python -m pytest tests/test_pipeline_gpu.py -x -q -k "world2 or contract_on_tiny" 2>&1 | tail -15 ) > gpurun_out/r6/t1_bench_tests.txt
tail -3 gpurun_out/r6/t1_bench_tests.txt
SKIP=1 bash tools/profile_round.sh sdxl 4 r6 2>&1 | tail -1 | cut -c1-300
bash tools/profile_round.sh sd15 1 r6 2>&1 | tail -1 | cut -c1-300
ls gpurun_out/profiles_r6 | head -50
