cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "$F" | tail -25 ) > gpurun_out/r6/gpu_suite_e.txt
tail -3 gpurun_out/r6/gpu_suite_e.txt
( timeout 300 python3 __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -3 ) > gpurun_out/r6/smoke_e.txt; cat gpurun_out/r6/smoke_e.txt
bash tools/profile_round.sh sd15 1 r6 2>&1 | tail -1 | cut -c1-300
( timeout 900 python3 bench.py 2>gpurun_out/r6/bench_e.err | tail -1 ) > gpurun_out/r6/bench_e.json; cut -c1-300 gpurun_out/r6/bench_e.json
