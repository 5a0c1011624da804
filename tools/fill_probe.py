"""Operand-fill ceiling of a CU (mlsd_probe_fill, csrc/hip/probe.hip): bytes per clock a CU pulls from L2 / MALL with nothing else to do,
by staging path (LDS-DMA / registers / registers + ds_write), working set and number of CUs engaged.  usage: python3 tools/fill_probe.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib
L = _lib.lib(); vp = _lib.vp
L.mlsd_probe_fill.argtypes = [ctypes.c_int, vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
K = 1280                                  # row stride of the panel in halfs (8192 x 1280 fp16 operand)
rows_total = 1 << 17
src = _lib.from_numpy(np.random.default_rng(0).integers(0, 255, size=rows_total * K * 2, dtype=np.uint8))
clocks = _lib.DeviceBuffer(256 * 8); sink = _lib.DeviceBuffer(64)
trips = 200
print(f"# {trips} trips x 64 KB per block, rows of 128 B at stride {2 * K} B; B/clk from the blocks' own s_memtime clocks (median), GB/s per CU and TB/s chip from HIP events")
for nblocks in (256, 128, 32, 8):
    for span_rows, what in ((4096, "panel 4096 rows = 0.5 MB of lines (L2 hits)"), (32768, "panel 32768 rows = 4 MB of lines per walk"),
                            (rows_total, "panel 131072 rows = 16 MB of lines (MALL / HBM)")):
        line = []
        for mode, name in ((0, "LDS-DMA"), (1, "registers"), (2, "registers + ds_write")):
            for _ in range(2):
                _lib.check(L.mlsd_probe_fill(mode, vp(src.ptr), 2 * K, span_rows, trips, nblocks, vp(clocks.ptr), vp(sink.ptr), None))
            L.mlsd_event_record(ev[0], None)
            _lib.check(L.mlsd_probe_fill(mode, vp(src.ptr), 2 * K, span_rows, trips, nblocks, vp(clocks.ptr), vp(sink.ptr), None))
            L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
            ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
            clk = np.median(clocks.download((256,), np.uint64)[:nblocks].astype(np.float64))
            byts = trips * 65536.0
            line.append(f"{name}: {byts / clk:5.1f} B/clk  {byts / (ms.value * 1e-3) / 1e9:6.1f} GB/s/CU  {byts * nblocks / (ms.value * 1e-3) / 1e12:5.2f} TB/s")
        print(f"{nblocks:4d} CUs, {what:48s} | " + " | ".join(line), flush=True)
