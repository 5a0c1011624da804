"""Per-kernel breakdown of one latent decode (KL-VAE or TAESD) plan (diagnostics, not the bench).
usage: python3 tools/quick_decode_perf.py <model> <latent_side> <n_batch> [tae] [flags]"""
import os, sys, ctypes, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import engine, _lib

model, lat, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tae = len(sys.argv) > 4 and sys.argv[4] == "tae"
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 16
L = _lib.lib()
dec = engine.Decoder(model, lat, lat, n, tae=tae)
_lib.lib().mlctx_set_flags(dec.ctx.h, flags)
info = dec.ctx.info()
print(f"ops={info.n_ops} flops={info.flops/1e12:.3f} TFLOP params={info.mem_params/2**20:.1f} MiB act={info.mem_compute/2**30:.2f} GiB")
x = np.random.default_rng(0).standard_normal((n, 4, lat, lat)).astype(np.float32)
dec.run(x); dec.run(x)
ev = [_lib.vp(), _lib.vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
K = 3
L.mlsd_event_record(ev[0], None)
for _ in range(K): dec.ctx.compute()
L.mlsd_event_record(ev[1], None)
L.mlsd_event_sync(ev[1])
ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
per = ms.value / K
print(f"decode {per:.2f} ms -> {info.flops/per/1e9:.1f} TFLOP/s")
ops, nb, tms = dec.ctx.op_list(), dec.ctx.op_bytes(), dec.ctx.profile_ops()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for (lab, fl), t, b in zip(ops, tms, nb):
    a = agg[lab]; a[0] += 1; a[1] += t; a[2] += fl; a[3] += b
for lab, (cnt, t, fl, b) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {lab:60s} n={cnt:3d} {t:8.2f} ms {fl/max(t,1e-9)/1e9:8.1f} TFLOP/s {b/max(t,1e-9)/1e6:8.1f} GB/s")
