R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
python3 - <<'PY'
import ctypes, sys, time
sys.path.insert(0, '.')
from mlimgsynth_amd import _lib
L = _lib.lib(); vp = _lib.vp
nb = 500 << 20
h = vp(); _lib.check(L.mlsd_host_alloc(ctypes.byref(h), ctypes.c_size_t(nb)), "host"); ctypes.memset(h.value, 1, nb)
d = _lib.DeviceBuffer(nb)
s = vp(); _lib.check(L.mlsd_stream_create(ctypes.byref(s)), "stream")
for rep in range(2):
    ts = []
    t0 = time.perf_counter()
    for i in range(6):
        _lib.check(L.mlsd_memcpy(vp(d.ptr), h, ctypes.c_size_t(nb), 0, s), "cpy"); ts.append((time.perf_counter() - t0) * 1e3)
    L.mlsd_stream_sync(s); tot = (time.perf_counter() - t0) * 1e3
    print("host time after each of 6 x 500 MiB hipMemcpyAsync calls (ms):", " ".join(f"{t:.2f}" for t in ts), "| all done at %.1f ms" % tot)
PY
