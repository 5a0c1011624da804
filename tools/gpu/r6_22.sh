# round 6, last call: GPU suite, smoke and the default bench invocation on the last commit
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "$F" | tail -25 ) > gpurun_out/r6/gpu_suite_d.txt
tail -3 gpurun_out/r6/gpu_suite_d.txt
( timeout 300 python3 __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -3 ) > gpurun_out/r6/smoke_d.txt; cat gpurun_out/r6/smoke_d.txt
( timeout 900 python3 bench.py 2>gpurun_out/r6/bench_d.err | tail -1 ) > gpurun_out/r6/bench_d.json; cut -c1-400 gpurun_out/r6/bench_d.json
