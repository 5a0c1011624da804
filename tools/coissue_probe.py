"""Do vector instructions run in the shadow of MFMAs on this part?  mlsd_probe_coissue: per iteration 8 slices of {one v_mfma_f32_32x32x16_f16 (32 matrix clocks), nv vector
instructions of one kind}, one or two waves per SIMD.  Prints shader clocks per slice for the MFMAs alone, the vector instructions alone, and both.
usage: python3 tools/coissue_probe.py [iters]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlimgsynth_amd import _lib
L = _lib.lib(); vp = _lib.vp
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
NB = 256
rng = np.random.default_rng(0)
src = _lib.from_numpy(rng.standard_normal(64 * 2048 * 8).astype(np.float16))
clk = _lib.DeviceBuffer(NB * 8); sink = _lib.DeviceBuffer(16)
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
def run(nt, mf, vk, nv):
    for _ in range(2): _lib.check(L.mlsd_probe_coissue(vp(src.ptr), iters, NB, nt, mf, vk, nv, vp(clk.ptr), vp(sink.ptr), None), "probe")
    L.mlsd_device_sync()
    L.mlsd_event_record(ev[0], None)
    for _ in range(4): _lib.check(L.mlsd_probe_coissue(vp(src.ptr), iters, NB, nt, mf, vk, nv, vp(clk.ptr), vp(sink.ptr), None), "probe")
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    c = float(np.median(clk.download((NB,), np.uint64)))
    return c / (iters * 8), c / (ms.value / 4 * 1e-3) / 1e9
names = {1: "v_fma_f32", 2: "v_exp_f32", 3: "v_pk_fma_f32", 4: "v_cvt_pk_f16_f32", 5: "v_max3_f32", 6: "v_pk_add_f32", 7: "v_add_f32", 8: "v_dot2_f32_f16", 9: "v_pk_mul_f32", 10: "v_pk_fma_f16", 11: "v_exp_f16", 12: "v_lshl_add_u64", 13: "v_mad_u64_u32", 14: "v_mul_lo_u32", 15: "v_add_u32", 16: "v_add_co+v_addc_co", 17: "v_add_f64"}
for nt in (256, 512):
    w = nt // 256
    m, g = run(nt, 1, 0, 0)
    print(f"{w} wave(s) per SIMD: MFMA 32x32x16 alone {m:6.1f} clocks per slice and wave ({g:.2f} GHz)")
    for vk in [int(v) for v in os.environ.get('COISSUE_KINDS', '').split(',') if v] or sorted(names):
        for nv in (3, 6):
            v, gv = run(nt, 0, vk, nv)
            b, gb = run(nt, 1, vk, nv)
            print(f"   {nv} x {names[vk]:17s}: alone {v:6.1f} clocks = {v / gv:5.1f} ns   with the MFMA {b:6.1f} = {b / gb:5.1f} ns   (MFMA alone {m / g:5.1f} ns; sum {m / g + v / gv:5.1f}, max {max(m / g, v / gv):5.1f})", flush=True)
