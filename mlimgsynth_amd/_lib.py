"""ctypes loader for libmlimgsynth_amd.so (the C-ABI library declared in include/*.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc for the
gfx950 kernels, gcc for the C host side).  There is deliberately NO fallback: if
the library is missing the import fails loudly (the product path never routes
through oracle/ or any CPU implementation).

torch is imported first when present so that the process uses ONE HIP runtime
(torch bundles libamdhip64.so with SONAME libamdhip64.so.7; loading ours after
torch makes the dynamic loader reuse it).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLSD_LIB_PATH") or os.path.join(_HERE, "lib", "libmlimgsynth_amd.so")     # (override: diagnostic builds, tools/gemm_trace.py)

try:  # plumbing only: device selection + torch.distributed live on the torch side
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

_lib = None


class MlsdError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the HIP extension is mandatory; there is no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _lib.mlsd_last_error.restype = ctypes.c_char_p
    return _lib


def last_error():
    return lib().mlsd_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    """Kernel-shim convention: 0 = ok, <0 = error."""
    if rc != 0:
        raise MlsdError(f"{what} failed (rc={rc}): {last_error()}")
    return rc


def check1(rc, what=""):
    """Host-side convention of the reference (ccommon.h TRY): >=1 ok, <0 error."""
    if rc < 0:
        raise MlsdError(f"{what} failed (rc={rc}): {last_error()}")
    return rc


vp = ctypes.c_void_p


class DeviceBuffer:
    """Raw device allocation through the C-ABI (hipMalloc)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = vp()
        check(lib().mlsd_malloc(ctypes.byref(p), ctypes.c_size_t(self.nbytes)), "mlsd_malloc")
        self.ptr = p.value

    def upload(self, arr, stream=None):
        import numpy as np
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().mlsd_memcpy(vp(self.ptr), arr.ctypes.data_as(vp), ctypes.c_size_t(arr.nbytes), 0, vp(stream)),
              "memcpy h2d")
        check(lib().mlsd_stream_sync(vp(stream)), "sync")
        return self

    def download(self, shape, dtype, stream=None, offset=0):
        import numpy as np
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes + offset <= self.nbytes
        check(lib().mlsd_memcpy(out.ctypes.data_as(vp), vp(self.ptr + offset), ctypes.c_size_t(out.nbytes), 1,
                                vp(stream)), "memcpy d2h")
        check(lib().mlsd_stream_sync(vp(stream)), "sync")
        return out

    def free(self):
        if self.ptr:
            lib().mlsd_free(vp(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def from_numpy(arr):
    import numpy as np
    arr = np.ascontiguousarray(arr)
    return DeviceBuffer(max(arr.nbytes, 16)).upload(arr)
