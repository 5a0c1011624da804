"""Pinned host -> device copy rate: one stream against two and four streams copying disjoint halves / quarters at the same time (the weight-streaming mode of
mlblock.c uploads a segment on ONE copy stream: 42.6 GB/s in the bench), and the chunk size.
usage: python3 tools/h2d_probe.py [MiB]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib
L = _lib.lib(); vp = _lib.vp
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nb = mib << 20
h = vp(); _lib.check(L.mlsd_host_alloc(ctypes.byref(h), ctypes.c_size_t(nb)), "host alloc")
ctypes.memset(h.value, 1, nb)
d = _lib.DeviceBuffer(nb)
streams = []
for _ in range(4):
    s = vp(); _lib.check(L.mlsd_stream_create(ctypes.byref(s)), "stream"); streams.append(s)


def run(ns, chunk):
    per = nb // ns
    def once():
        for k in range(ns):
            off = k * per
            while off < (k + 1) * per:
                n = min(chunk, (k + 1) * per - off)
                _lib.check(L.mlsd_memcpy(vp(d.ptr + off), vp(h.value + off), ctypes.c_size_t(n), 0, streams[k]), "memcpy")
                off += n
        for k in range(ns): L.mlsd_stream_sync(streams[k])
    once()
    t0 = time.perf_counter()
    for _ in range(3): once()
    return nb * 3 / (time.perf_counter() - t0) / 1e9


for ns in (1, 2, 4):
    for chunk in (8 << 20, 64 << 20, nb):
        print(f"{ns} stream(s), chunks of {chunk >> 20:5d} MiB: {run(ns, chunk):6.1f} GB/s", flush=True)
