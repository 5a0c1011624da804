cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for i in 1 2 3 4 5 6 7 8; do ( timeout 900 python3 tools/soak_r5.py 4000 5 2>&1 | grep -v "$F" ) > gpurun_out/r6/soak_f$i.txt; grep "MISMATCH\|COUNTERS\|soak" gpurun_out/r6/soak_f$i.txt | head -4 | cut -c1-160; done
for i in 1 2 3 4; do ( timeout 900 python3 tools/soak_r5.py 4000 5 pp 2>&1 | grep -v "$F" ) > gpurun_out/r6/soak_fpp$i.txt; grep "MISMATCH\|COUNTERS\|soak" gpurun_out/r6/soak_fpp$i.txt | head -4 | cut -c1-160; done
( timeout 900 python3 tools/soak_streamk_alt.py 2>&1 | grep -v "$F" | tail -4 ) 
( timeout 1200 python -m pytest tests/test_determinism_gpu.py -q 2>&1 | grep -v "$F" | tail -2 )
