R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R; export PYTHONPATH=$R
python3 - <<'PY'
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np
from mlimgsynth_amd import engine, _lib
L = _lib.lib(); vp = _lib.vp
un = engine.Unet("sdxl", 128, 128, 8, stream_weights_mib=512)
h = vp(); _lib.check(L.mlsd_host_alloc(ctypes.byref(h), ctypes.c_size_t(1 << 20)), "host")
d = _lib.DeviceBuffer(1 << 20)
def period(nbytes, kind):
    for _ in range(3): un.ctx.compute()
    un.ctx.sync(); t0 = time.perf_counter()
    for _ in range(10):
        if nbytes: _lib.check(L.mlsd_memcpy(vp(d.ptr), h if kind == 0 else vp(d.ptr + (1 << 19)), ctypes.c_size_t(nbytes), kind, None), "cpy")   # on the plan's (NULL) stream
        un.ctx.compute()
    un.ctx.sync(); return (time.perf_counter() - t0) * 100
print("no extra copy          : %.2f ms" % period(0, 0))
print("+ 64 B H2D per eval    : %.2f ms" % period(64, 0))
print("+ 256 KB H2D per eval  : %.2f ms" % period(256 << 10, 0))
print("+ 256 KB D2D per eval  : %.2f ms" % period(256 << 10, 2))
PY
