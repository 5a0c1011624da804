cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for i in 1 2 3; do ( timeout 900 python3 tools/soak_r5.py 3000 20 2>&1 | grep -v "$F" ) > gpurun_out/r6/soak_d$i.txt; grep -c MISMATCH gpurun_out/r6/soak_d$i.txt; grep "MISMATCH\|COUNTERS\|soak" gpurun_out/r6/soak_d$i.txt | head -12 | cut -c1-200; done
