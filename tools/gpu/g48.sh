R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "stream or split" 2>&1 | tail -3
timeout 600 python3 tools/soak_stream.py 200 6 2>&1 | grep -v amdgpu
python3 - <<'PY'
import sys, time
sys.path.insert(0, '.')
from mlimgsynth_amd import engine
for mib in (512, 384, 640):
    un = engine.Unet("sdxl", 128, 128, 8, stream_weights_mib=mib)
    for _ in range(3): un.ctx.compute()
    un.ctx.sync(); t0 = time.perf_counter()
    for _ in range(10): un.ctx.compute()
    un.ctx.sync(); print("slab %d MiB: %d segments, streamed eval period %.2f ms" % (mib, un.ctx.streaming_info()[0], (time.perf_counter() - t0) * 100), flush=True)
    un.ctx.destroy()
PY
