"""What the matrix pipes sustain with nothing else running (mlsd_probe_mfma_rate: 256 blocks x 8 waves issuing independent v_mfma_f32_16x16x32_f16 on register operands):
TFLOP/s by wall clock, the shader clock the part holds meanwhile, and pipe utilisation per clock -- for random operands and for zeros (the power of an MFMA depends on its data).
usage: python3 tools/mfma_rate.py [iters] [launches]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlimgsynth_amd import _lib
L = _lib.lib(); vp = _lib.vp
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 12
NB = 256
rng = np.random.default_rng(0)
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
clk = _lib.DeviceBuffer(NB * 8); sink = _lib.DeviceBuffer(16)
for name, data in (("random normal fp16 operands", rng.standard_normal(64 * 2048 * 8).astype(np.float16)),
                   ("small-range operands (|x| < 2^-6)", (rng.standard_normal(64 * 2048 * 8) * 2.0 ** -8).astype(np.float16)),
                   ("all-zero operands", np.zeros(64 * 2048 * 8, np.float16))):
    src = _lib.from_numpy(data)
    for _ in range(2): _lib.check(L.mlsd_probe_mfma_rate(vp(src.ptr), iters, NB, vp(clk.ptr), vp(sink.ptr), None), "probe")
    L.mlsd_device_sync()
    L.mlsd_event_record(ev[0], None)
    for _ in range(launches): _lib.check(L.mlsd_probe_mfma_rate(vp(src.ptr), iters, NB, vp(clk.ptr), vp(sink.ptr), None), "probe")
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    per = ms.value / launches * 1e-3
    W = 4 if os.environ.get('MLSD_PROBE_ONE_WAVE', '0') not in ('', '0') else 8
    flop = NB * W * iters * 8 * 16384.0
    c = float(np.median(clk.download((NB,), np.uint64)))
    # a SIMD runs 2 waves x 8 MFMAs per iteration; one 16x16x32 fp16 MFMA = 16 pipe clocks at the peak rate (1024 FLOP / clock / SIMD)
    print(f"{name:36s}: {flop / per / 1e12:7.1f} TFLOP/s over {per * 1e3:.2f} ms launches | loop {c:.0f} clocks -> {c / per / 1e9:.3f} GHz held | "
          f"pipe busy per clock {iters * (W // 4) * 8 * 16 / c:.3f} ({W // 4} wave(s) per SIMD)", flush=True)
