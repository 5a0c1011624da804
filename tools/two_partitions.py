"""Experiment: two independent half-batch UNet plans side by side on two CU-masked streams (each half of the chip) against
the one full-batch plan on the whole chip.  The single-round GEMMs, the HBM-bound norm kernels and every launch's prologue /
epilogue leave the matrix pipes idle for part of the time; two desynchronised half-chip streams may fill those gaps.
usage: python3 tools/two_partitions.py [model] [latent] [N] [reps] [mask_mode ...]
  mask_mode: half = mask bits [0,128) / [128,256);  parity = even / odd bits;  none = two unmasked streams"""
import ctypes, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, engine
L = _lib.lib(); vp = _lib.vp
model = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
lat = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
modes = sys.argv[5:] or ["half", "parity", "none"]
L.mlsd_gemm_set_cus.argtypes = [ctypes.c_int]
FLAGS = int(os.environ.get('PLAN_FLAGS', '0'))      # 8 = replay each plan as a hipGraph


def stream(mask=None):
    s = vp()
    if mask is None:
        _lib.check(L.mlsd_stream_create(ctypes.byref(s)), "stream")
    else:
        m = (ctypes.c_uint32 * 8)(*mask)
        _lib.check(L.mlsd_stream_create_masked(ctypes.byref(s), m, 8), "masked stream")
    return s.value


def masks(mode):
    if mode == "half":
        return [0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4
    if mode == "parity":
        return [0x55555555] * 8, [0xAAAAAAAA] * 8
    if mode == "quad":      # bits 0-3 of every 8 / bits 4-7 of every 8
        return [0x0F0F0F0F] * 8, [0xF0F0F0F0] * 8
    return None, None


def census(s, n=512):
    buf = _lib.DeviceBuffer(n * 4)
    L.mlsd_cu_census.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
    _lib.check(L.mlsd_cu_census(vp(buf.ptr), n, 200000, vp(s)), "census")
    v = buf.download((n,), np.uint32, stream=s)
    xcc = v >> 16
    cus = len(set((v & 0xFFFFFF00).tolist()))      # (XCC, SE, SH, CU) of HW_ID
    return sorted(set(xcc.tolist())), cus


def timed(fn, sync):
    for _ in range(2): fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3


s0 = stream()
full = engine.Unet(model, lat, lat, N, stream=s0, flags=FLAGS)
t_full = timed(lambda: full.ctx.compute(), lambda: full.ctx.sync())
print(f"{model} latent {lat}: whole chip, one plan of batch {N}: {t_full:.2f} ms per evaluation", flush=True)
for mode in modes:
    ma, mb = masks(mode)
    sa, sb = stream(ma), stream(mb)
    if ma is not None:
        print(f"[{mode}] stream A: XCCs {census(sa)}, stream B: XCCs {census(sb)}", flush=True)
    L.mlsd_gemm_set_cus(128 if ma is not None else 256)
    ha, hb = engine.Unet(model, lat, lat, N // 2, stream=sa, flags=FLAGS), engine.Unet(model, lat, lat, N // 2, stream=sb, flags=FLAGS)
    t_half = timed(lambda: ha.ctx.compute(), lambda: ha.ctx.sync())
    # one host thread per stream (ctypes releases the GIL inside the library): the two streams drift apart by themselves
    for delay_ms in (0.0, 15.0, 30.0):
        def run(u, d):
            if d: time.sleep(d * 1e-3)
            for _ in range(reps): u.ctx.compute()
            u.ctx.sync()
        for u in (ha, hb):
            u.ctx.compute(); u.ctx.sync()
        th = [threading.Thread(target=run, args=(ha, 0.0)), threading.Thread(target=run, args=(hb, delay_ms))]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        el = (time.perf_counter() - t0 - delay_ms * 1e-3 * 0) / reps * 1e3
        print(f"[{mode}] half plan alone on its partition: {t_half:.2f} ms | two partitions, start offset {delay_ms:.0f} ms: "
              f"{el:.2f} ms per {N} inputs (incl. the offset once: {delay_ms / reps:.2f} ms) vs {t_full:.2f}", flush=True)
    L.mlsd_gemm_set_cus(256)
    del ha, hb
print("tile-table misses:", L.mlctx_tune_misses())
