R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "stream or split" 2>&1 | tail -4
timeout 300 python3 tools/wstream_trace.py sdxl 128 8 512 > $O/r4_wstream_trace3.txt 2>&1; echo rc=$?; grep wstream $O/r4_wstream_trace3.txt | tail -10
timeout 1200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_stream.json 2> $O/bench_stream.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4s/bench_stream.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "eval", d.get("unet_eval_ms"))
for k in ("sd15","sdxl_tae","sdxl_b8","sdxl_tae_split"): print(k, {kk:d[k][kk] for kk in d[k] if kk in ('value','ms_per_step','unet_eval_ms','weight_streaming')})
PY
