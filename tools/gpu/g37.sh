R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_probe.json 2> $O/bench_probe.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4s/bench_probe.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"])
PY
timeout 600 python3 -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "bench_line" 2>&1 | tail -3
