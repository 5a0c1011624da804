R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "stream" 2>&1 | tail -4
timeout 1200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_stream.json 2> $O/bench_stream.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4s/bench_stream.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "eval", d.get("unet_eval_ms"))
for k,v in d.get("extras",{}).items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("unet_eval_ms"), v.get("weight_streaming"))
PY
