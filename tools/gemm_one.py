"""Run one GEMM shape/variant a few times (for rocprofv3 --pmc runs)."""
import sys
sys.path.insert(0, ".")
from tools import gemm_bench_lib as g
v, M, N, K = [int(x) for x in sys.argv[1:5]]
print(g.time_gemm(M, N, K, v, reps=5))
