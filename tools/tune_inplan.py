"""IN-PLAN tile tuning (mlctx_tune_inplan): every candidate tile of every GEMM shape of a plan timed where it runs (cold weights, HBM residuals, the
plan's clocks), against the compiled-in table's choice; a challenger must win by 3 %.  Writes the changed shapes as lines for
mlimgsynth_amd/csrc/host/tune_table.inc (replace the lines of the same key) and times the evaluation before / after.
usage (GPU box): python3 tools/tune_inplan.py <out.inc> [unet:sdxl:128:8] [unet:sd1:64:2] [vae:sdxl:128:4] ...   (default: the three bench plans)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, engine
L = _lib.lib(); vp = _lib.vp
L.mlctx_tune_inplan.argtypes = [vp, ctypes.c_int]
out = sys.argv[1]
plans = sys.argv[2:] or ["unet:sdxl:128:8", "unet:sd1:64:2", "vae:sdxl:128:4", "vae:sd1:64:1"]
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))


def eval_ms(ctx, k=5):
    for _ in range(2): ctx.compute()
    ctx.sync()
    L.mlsd_event_record(ev[0], None)
    for _ in range(k): ctx.compute()
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / k


for spec in plans:
    kind, model, lat, n = spec.split(":")
    lw, lh = (int(v) for v in lat.split("x")) if "x" in lat else (int(lat), int(lat))      # latent size: 128 or 128x96
    n = int(n)
    obj = engine.Unet(model, lw, lh, n) if kind == "unet" else engine.Decoder(model, lw, lh, n, tae=(kind == "tae"))
    ctx = obj.ctx
    t0 = eval_ms(ctx)
    nch = L.mlctx_tune_inplan(ctx.h, 3)
    if nch < 0: raise SystemExit("mlctx_tune_inplan failed: " + _lib.last_error())
    t1 = eval_ms(ctx)
    print(f"{spec}: {nch} shapes changed; evaluation {t0:.3f} -> {t1:.3f} ms", flush=True)
    ctx.destroy()
n = L.mlsd_tune_dump(out.encode())
print(f"{n} table lines written to {out}")
