R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4w; mkdir -p $O; cd $R; export PYTHONPATH=$R
bash tools/gpu/g53.sh 2>&1 | grep -E "total|gap to the next +[1-9][0-9]?\.[0-9]+ ms" | grep -v "gap to the next   0\.0" | tail -8
timeout 1200 python3 -m pytest tests/test_pipeline_gpu.py tests/test_sampler_gpu.py tests/test_determinism_gpu.py tests/test_api_gpu.py tests/test_golden_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 1200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench3.json 2> $O/bench3.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4w/bench3.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "eval", d.get("unet_eval_ms"))
for k in ("sd15","sdxl_tae","sdxl_b8","sdxl_tae_split"): print(k, {kk:d[k][kk] for kk in d[k] if kk in ('value','ms_per_step','unet_eval_ms')}, d[k].get("weight_streaming",{}).get("h2d_gb_per_s_sustained"))
PY
