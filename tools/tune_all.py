"""OFFLINE tile tuning: times the GEMM/conv tile variants of every shape of the listed workloads on the GPU and writes
the winners as lines of mlimgsynth_amd/csrc/host/tune_table.inc (the table the library compiles in, so that run-time tile
selection is a pure function of the shape: mlblock.c "GEMM tile selection").

usage (GPU box):  python3 tools/tune_all.py gpurun_out/tune_table.inc [--fresh] [--only sdxl,sd15,...]
Then, in the build container: copy the file over mlimgsynth_amd/csrc/host/tune_table.inc (or append the new lines) and rebuild.
Without --fresh only shapes the compiled-in table does not hold are timed (the output then holds just the new lines).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1]
fresh = "--fresh" in sys.argv
only = None
for i, a in enumerate(sys.argv):
    if a == "--only":
        only = set(sys.argv[i + 1].split(","))
if fresh:
    os.environ["MLSD_TUNE_IGNORE_TABLE"] = "1"
os.environ["MLSD_AUTOTUNE"] = "1"

import numpy as np  # noqa: E402
from mlimgsynth_amd import _lib, engine, text  # noqa: E402

L = _lib.lib()


def gen(model, w, h, b, tae=False):
    t0 = time.time()
    g = engine.Generator(model, w, h, b, n_step=1, use_tae=tae)
    P = g.P
    rng = np.random.default_rng(0)
    cond = rng.standard_normal((77, P.n_ctx)).astype(np.float32)
    lab = rng.standard_normal(P.ch_adm_in).astype(np.float32) if P.ch_adm_in else None
    g.set_cond(cond, lab, cond, lab)
    g.generate(list(range(b)), want_latents=False, want_images=False)   # first compute of both plans: timing pass
    g.destroy()
    print(f"tuned generator {model} {w}x{h} b{b} tae={tae} in {time.time() - t0:.1f}s", flush=True)


def unet(model, lat, n):
    t0 = time.time()
    un = engine.Unet(model, lat, lat, n)
    un.ctx.compute(); un.ctx.sync(); un.ctx.destroy()
    print(f"tuned unet {model} {lat} n{n} in {time.time() - t0:.1f}s", flush=True)


def dec(model, lat, n, tae=False):
    t0 = time.time()
    d = engine.Decoder(model, lat, lat, n, tae=tae)
    d.ctx.compute(); d.ctx.sync(); d.ctx.destroy()
    print(f"tuned decoder {model} {lat} n{n} tae={tae} in {time.time() - t0:.1f}s", flush=True)


def textc(model, w, h):
    tc = text.TextConditioner(model, w, h)
    tc.encode_pair(np.arange(8, dtype=np.int32), ())
    print(f"tuned text towers {model}", flush=True)


JOBS = {
    # bench workloads (BASELINE configs[1..4]) and the batch sizes around them
    "sdxl": lambda: [gen("sdxl", 1024, 1024, b) for b in (4, 8, 2, 1)] + [gen("sdxl", 1024, 1024, 4, tae=True), textc("sdxl", 1024, 1024)],
    "sd15": lambda: [gen("sd1", 512, 512, b) for b in (1, 2, 4)] + [gen("sd1", 512, 512, 1, tae=True), textc("sd1", 512, 512)],
    # shapes of the parity tests (tests/test_unet_gpu.py, test_pipeline_gpu.py, test_golden_gpu.py)
    "tests": lambda: [unet("sdxl", 128, 1), unet("sdxl", 32, 2), unet("sdxl", 16, 2), unet("sd1", 16, 2), unet("sd1", 64, 1),
                      dec("sd1", 64, 1), dec("sdxl", 128, 1), dec("sdxl", 128, 1, tae=True), dec("sd1", 8, 1)],
}
for name, job in JOBS.items():
    if only and name not in only:
        continue
    job()
L.mlsd_tune_dump.argtypes = [__import__("ctypes").c_char_p]
n = L.mlsd_tune_dump(out.encode())
print(f"wrote {n} table lines to {out}")
