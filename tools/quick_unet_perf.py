"""Quick per-evaluation timing of the UNet plan + per-kernel breakdown (diagnostics, not the bench)."""
import sys, time, ctypes, collections
import numpy as np
sys.path.insert(0, ".")
from mlimgsynth_amd import engine, _lib

model, lat, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
import os
_lib.lib().mlsd_gemm_set_panel(int(os.environ.get('GEMM_MODE', '8')))
t0 = time.time()
un = engine.Unet(model, lat, lat, n, flags=flags)
info = un.ctx.info()
print(f"build+synth {time.time()-t0:.1f}s ops={info.n_ops} flops/eval={info.flops/1e12:.3f} TFLOP params={info.mem_params/2**30:.2f} GiB act={info.mem_compute/2**30:.2f} GiB")
L = _lib.lib()
for _ in range(2):
    un.ctx.compute()
un.ctx.sync()
ev = [_lib.vp(), _lib.vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
K = 5
L.mlsd_event_record(ev[0], None)
for _ in range(K): un.ctx.compute()
L.mlsd_event_record(ev[1], None)
L.mlsd_event_sync(ev[1])
ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
per = ms.value / K
print(f"eval {per:.2f} ms  -> {info.flops/per/1e9:.1f} TFLOP/s ({info.flops/per/1e9/2500*100:.1f}% of 2.5 PF)")
ops = un.ctx.op_list()
tms = un.ctx.profile_ops()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (lab, fl), t in zip(ops, tms):
    a = agg[lab]; a[0] += 1; a[1] += t; a[2] += fl
print("sum of per-op times %.2f ms" % tms.sum())
for lab, (cnt, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {lab:40s} n={cnt:4d} {t:8.2f} ms  {fl/1e12:7.3f} TFLOP  {fl/max(t,1e-9)/1e9:8.1f} TFLOP/s")
