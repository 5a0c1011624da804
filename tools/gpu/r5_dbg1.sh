cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
python3 - <<'PY'
import os
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("cpu.max", e)
PY
( time OMP_WAIT_POLICY=PASSIVE OMP_PROC_BIND=false timeout 120 python3 bench.py --cpu-child sd1 512 512 7.0 40 64 60 2.515e12 ) 2>&1 | grep -E "CPU|real" 
( time OMP_WAIT_POLICY=PASSIVE OMP_PROC_BIND=false timeout 100 python3 bench.py --cpu-child sd1 512 512 7.0 40 32 60 2.515e12 ) 2>&1 | grep -E "CPU|real" 
for e in "X=1" "MLSD_NO_NEAREST_TILE=1" "MLSD_TT=0" "MLSD_TT_LN=0"; do echo "== $e"; env $e timeout 300 python -m pytest tests/test_golden_gpu.py -x -q -s -k "unet_sdxl_16" 2>&1 | grep -E "^unet_sdxl_16|passed|failed" | head -3; done
