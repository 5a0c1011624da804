# round 6, call 15: the final tree -- whole GPU suite, alternating-operand soak, smoke, the round's profile set (rocprofv3 kernel stats, PMC traffic / MFMA, shape tables, bench lines)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "$F" | tail -25 ) > gpurun_out/r6/gpu_suite_c.txt
tail -4 gpurun_out/r6/gpu_suite_c.txt
( timeout 900 python3 tools/soak_r5.py 3000 100 2>&1 | grep -v "$F" | tail -12 ) > gpurun_out/r6/soak_c.txt
tail -3 gpurun_out/r6/soak_c.txt
( timeout 300 python3 __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -3 ) > gpurun_out/r6/smoke_c.txt; cat gpurun_out/r6/smoke_c.txt
bash tools/profile_round.sh sdxl 4 r6 2>&1 | tail -1 | cut -c1-400
bash tools/profile_round.sh sd15 1 r6 2>&1 | tail -1 | cut -c1-300
ls gpurun_out/profiles_r6 | head -60
( timeout 900 python3 bench.py 2>gpurun_out/r6/bench_c.err | tail -1 ) > gpurun_out/r6/bench_c.json; cut -c1-700 gpurun_out/r6/bench_c.json
