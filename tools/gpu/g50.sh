R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 1200 python3 -m pytest tests/test_pipeline_gpu.py tests/test_sampler_gpu.py tests/test_determinism_gpu.py tests/test_api_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
timeout 1200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4v/bench2.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "eval", d.get("unet_eval_ms"))
for k in ("sd15","sdxl_tae","sdxl_b8","sdxl_tae_split"): print(k, {kk:d[k][kk] for kk in d[k] if kk in ('value','ms_per_step','unet_eval_ms')}, d[k].get("weight_streaming",{}).get("h2d_gb_per_s_sustained"))
PY
