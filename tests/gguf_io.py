"""Minimal GGUF v2/v3 WRITER for tests (numpy only) + independent block quantisers / dequantisers.

Layout written (the published GGUF container, the same one src/ccompute/tensorstore_gguf.c reads):
  magic "GGUF" | u32 version | u64 n_tensor | u64 n_meta | n_meta x {string key, u32 type, value} |
  n_tensor x {string name, u32 n_dim, u64 dims[n_dim] (fastest first), u32 ggml_type, u64 offset} | pad to 32 | data
Strings are u64 length + bytes.  Tensor offsets are relative to the start of the (32-byte aligned) data section."""
import struct

import numpy as np

GGML = {"F32": 0, "F16": 1, "Q4_0": 2, "Q4_1": 3, "Q5_0": 6, "Q5_1": 7, "Q8_0": 8, "BF16": 30}
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)


def _s(x):
    b = x.encode() if isinstance(x, str) else x
    return struct.pack("<Q", len(b)) + b


def _val(t, v):
    fmt = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i", T_F32: "<f", T_BOOL: "<B", T_U64: "<Q", T_I64: "<q", T_F64: "<d"}
    if t == T_STR:
        return _s(v)
    if t == T_ARR:
        et, items = v
        return struct.pack("<IQ", et, len(items)) + b"".join(_val(et, i) for i in items)
    return struct.pack(fmt[t], v)


def quantise(x, kind):
    """x: float32 [..., 32*k] -> (bytes, dequantised float32 array).  The expected values are computed HERE, from the integer
    codes this function chose, by the block formulas; they do not pass through the library under test."""
    x = np.asarray(x, np.float32)
    blocks = x.reshape(-1, 32)
    out, deq = [], np.empty_like(blocks)
    for i, b in enumerate(blocks):
        if kind == "Q8_0":
            d = np.float16(np.abs(b).max() / 127.0) if np.abs(b).max() > 0 else np.float16(0)
            q = np.clip(np.round(b / np.float32(d)) if d != 0 else np.zeros(32), -128, 127).astype(np.int8)
            out.append(d.tobytes() + q.tobytes())
            deq[i] = np.float32(d) * q.astype(np.float32)
            continue
        lo, hi = b.min(), b.max()
        nlev = 16 if kind in ("Q4_0", "Q4_1") else 32
        if kind in ("Q4_1", "Q5_1"):
            d = np.float16((hi - lo) / (nlev - 1)) if hi > lo else np.float16(0)
            m = np.float16(lo)
            q = np.clip(np.round((b - np.float32(m)) / np.float32(d)) if d != 0 else np.zeros(32), 0, nlev - 1).astype(np.uint8)
            deq[i] = np.float32(d) * q.astype(np.float32) + np.float32(m)
            head = d.tobytes() + m.tobytes()
        else:
            amax = b[np.argmax(np.abs(b))]
            d = np.float16(amax / -(nlev // 2)) if amax != 0 else np.float16(0)
            q = np.clip(np.round(b / np.float32(d)) + nlev // 2 if d != 0 else np.full(32, nlev // 2), 0, nlev - 1).astype(np.uint8)
            deq[i] = np.float32(d) * (q.astype(np.float32) - nlev // 2)
            head = d.tobytes()
        qs = ((q[:16] & 15) | ((q[16:] & 15) << 4)).astype(np.uint8)
        if nlev == 32:
            qh = 0
            for j in range(16):
                qh |= int(q[j] >> 4) << j
                qh |= int(q[j + 16] >> 4) << (j + 16)
            head += struct.pack("<I", qh)
        out.append(head + qs.tobytes())
    return b"".join(out), deq.reshape(x.shape)


def to_bf16_bytes(x):
    u = np.asarray(x, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16).tobytes()


def write(path, tensors, meta=(), version=3):
    """tensors: list of (name, array float32 in numpy (row-major) shape, kind).  Returns {name: expected float32 values}."""
    infos, blobs, expect, off = [], [], {}, 0
    for name, arr, kind in tensors:
        arr = np.ascontiguousarray(arr, np.float32)
        if kind == "F32":
            data, want = arr.tobytes(), arr
        elif kind == "F16":
            data, want = arr.astype(np.float16).tobytes(), arr.astype(np.float16).astype(np.float32)
        elif kind == "BF16":
            data = to_bf16_bytes(arr)
            want = (np.frombuffer(data, np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(arr.shape)
        else:
            data, want = quantise(arr, kind)
        dims = list(arr.shape[::-1])                       # fastest first
        infos.append(_s(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", GGML[kind], off))
        pad = (-len(data)) % 32
        blobs.append(data + b"\0" * pad)
        off += len(data) + pad
        expect[name] = want
    head = b"GGUF" + struct.pack("<IQQ", version, len(tensors), len(meta))
    for k, t, v in meta:
        head += _s(k) + struct.pack("<I", t) + _val(t, v)
    head += b"".join(infos)
    head += b"\0" * ((-len(head)) % 32)
    with open(path, "wb") as f:
        f.write(head + b"".join(blobs))
    return expect
