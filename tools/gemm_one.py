"""Run one GEMM shape/variant a few times (for rocprofv3 --pmc runs and loop ablations)."""
import sys
sys.path.insert(0, ".")
from tools import gemm_bench_lib as g
v, M, N, K = [int(x) for x in sys.argv[1:5]]
dbg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
g.L.mlsd_gemm_set_debug(dbg)
print("variant", v, "dbg", dbg, g.time_gemm(M, N, K, v, reps=10))
