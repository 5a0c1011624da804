# round 6, call 9: whole GPU suite on the current tree (self-tagged LayerNorm records, xattn fill rule), alternating-operand soak, smoke, bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "$F" | tail -25 ) > gpurun_out/r6/gpu_suite_b.txt
tail -6 gpurun_out/r6/gpu_suite_b.txt
( timeout 900 python3 tools/soak_r5.py 3000 100 2>&1 | grep -v "$F" | tail -12 ) > gpurun_out/r6/soak_b.txt
tail -5 gpurun_out/r6/soak_b.txt
( timeout 300 python3 __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -3 ) > gpurun_out/r6/smoke_b.txt; cat gpurun_out/r6/smoke_b.txt
( timeout 900 python3 bench.py 2>gpurun_out/r6/bench_b.err | tail -1 ) > gpurun_out/r6/bench_b.json; cut -c1-900 gpurun_out/r6/bench_b.json
