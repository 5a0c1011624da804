"""In-plan A/B of the attention kernel choices (which kernel takes the 1024- / 4096-token launches, query blocks per workgroup of the one-pass cross attention, row sums on the
VALU or the matrix pipe): one SDXL b4 UNet plan, every setting timed on whole evaluations, two rounds.  usage: python3 tools/attn_inplan.py"""
import sys, ctypes
sys.path.insert(0, ".")
from mlimgsynth_amd import engine, _lib
L = _lib.lib(); vp = _lib.vp
un = engine.Unet("sdxl", 128, 128, 8)
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
def ev_ms(k=5):
    for _ in range(2): un.ctx.compute()
    un.ctx.sync()
    L.mlsd_event_record(ev[0], None)
    for _ in range(k): un.ctx.compute()
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)); return ms.value / k
def reset():
    L.mlsd_attention_force_old(0); L.mlsd_attention_x2_min_tq(2048); L.mlsd_attention_tk96(1, 0); L.mlsd_attention_vsum(1)
L.mlsd_attention_tk96.argtypes = [ctypes.c_int, ctypes.c_int]
for rnd in range(2):
    for name, f in [("default", lambda: None), ("64-row kernel also at 1024 tokens", lambda: L.mlsd_attention_x2_min_tq(1024)),
                    ("32-row kernel everywhere", lambda: L.mlsd_attention_force_old(1)), ("one-pass qb 1", lambda: L.mlsd_attention_tk96(1, 1)),
                    ("one-pass qb 2", lambda: L.mlsd_attention_tk96(1, 2)), ("one-pass qb 4", lambda: L.mlsd_attention_tk96(1, 4)),
                    ("one-pass off", lambda: L.mlsd_attention_tk96(0, 0)), ("row sums on the matrix pipe", lambda: L.mlsd_attention_vsum(0))]:
        reset(); f()
        print(f"{name:40s} eval {ev_ms():.3f} ms", flush=True)
