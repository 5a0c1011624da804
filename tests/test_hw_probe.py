"""gfx950 lane-map probes: pin the MFMA f16 fragment layouts and the transposed
LDS read the attention / GEMM kernels rely on, with exact integer data."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _c_map(raw):
    """raw [64 lanes][16 regs] -> C[32][32] using col=lane&31, row=(reg&3)+8*(reg>>2)+4*(lane>>5)."""
    C = np.zeros((32, 32), np.float32)
    for l in range(64):
        for r in range(16):
            C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = raw[l, r]
    return C


def test_runtime_shared_with_torch():
    import torch
    from mlimgsynth_amd import _lib
    L = _lib.lib()
    assert L.mlsd_device_count() >= 1
    # a torch allocation is usable by our kernels and vice versa (one HIP runtime in the process)
    x = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
    y = torch.zeros_like(x)
    _lib.check(L.mlsd_probe_copy(_lib.vp(x.data_ptr()), _lib.vp(y.data_ptr()), ctypes.c_size_t(x.numel() * 4), None))
    _lib.check(L.mlsd_device_sync())
    assert torch.equal(x, y)


def test_mfma_32x32x16_f16_layout():
    from mlimgsynth_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    A = rng.integers(-4, 5, (32, 16)).astype(np.float16)
    B = rng.integers(-4, 5, (16, 32)).astype(np.float16)
    dA, dB = _lib.from_numpy(A), _lib.from_numpy(B)
    dC = _lib.DeviceBuffer(64 * 16 * 4)
    _lib.check(L.mlsd_probe_mfma_raw(_lib.vp(dA.ptr), _lib.vp(dB.ptr), _lib.vp(dC.ptr), None))
    raw = dC.download((64, 16), np.float32)
    os.makedirs(OUT, exist_ok=True)
    np.save(os.path.join(OUT, "probe_mfma_raw.npy"), raw)
    ref = A.astype(np.float32) @ B.astype(np.float32)
    assert np.array_equal(_c_map(raw), ref)


def test_mfma_accumulator_as_b_operand():
    from mlimgsynth_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(1)
    A1 = rng.integers(-3, 4, (32, 16)).astype(np.float16)
    B1 = rng.integers(-3, 4, (16, 32)).astype(np.float16)
    A2 = rng.integers(-3, 4, (32, 32)).astype(np.float16)
    d = [_lib.from_numpy(a) for a in (A1, B1, A2)]
    dY = _lib.DeviceBuffer(64 * 16 * 4)
    _lib.check(L.mlsd_probe_mfma_chain(_lib.vp(d[0].ptr), _lib.vp(d[1].ptr), _lib.vp(d[2].ptr), _lib.vp(dY.ptr), None))
    raw = dY.download((64, 16), np.float32)
    np.save(os.path.join(OUT, "probe_mfma_chain.npy"), raw)
    X = A1.astype(np.float32) @ B1.astype(np.float32)
    ref = A2.astype(np.float32) @ X
    assert np.array_equal(_c_map(raw), ref)


def test_ds_read_tr16_b64():
    from mlimgsynth_amd import _lib
    L = _lib.lib()
    T = (np.arange(8)[:, None] * 64 + np.arange(32)[None, :]).astype(np.float16)  # value = row*64+col (exact)
    dT = _lib.from_numpy(T)
    dO = _lib.DeviceBuffer(64 * 4 * 4)
    _lib.check(L.mlsd_probe_tr_read(_lib.vp(dT.ptr), 32, _lib.vp(dO.ptr), None))
    out = dO.download((64, 4), np.float32)
    np.save(os.path.join(OUT, "probe_tr_read.npy"), out)
    # expectation: lane L (g=L>>4, i=L&15) receives T[4*(g>>1) + j][16*(g&1) + i], j=0..3
    exp = np.zeros((64, 4), np.float32)
    for l in range(64):
        g, i = l >> 4, l & 15
        for j in range(4):
            exp[l, j] = T[4 * (g >> 1) + j, 16 * (g & 1) + i]
    assert np.array_equal(out, exp), (out[:20], exp[:20])


def test_hbm_copy_bandwidth_sane():
    """A plain float4 copy should stream at TB/s on HBM3E (guide: ~6.3 TB/s achievable)."""
    from mlimgsynth_amd import _lib
    L = _lib.lib()
    n = 1 << 30
    a, b = _lib.DeviceBuffer(n), _lib.DeviceBuffer(n)
    ev = [_lib.vp(), _lib.vp()]
    for e in ev:
        _lib.check(L.mlsd_event_create(ctypes.byref(e)))
    for _ in range(2):
        _lib.check(L.mlsd_probe_copy(_lib.vp(a.ptr), _lib.vp(b.ptr), ctypes.c_size_t(n), None))
    _lib.check(L.mlsd_event_record(ev[0], None))
    for _ in range(5):
        _lib.check(L.mlsd_probe_copy(_lib.vp(a.ptr), _lib.vp(b.ptr), ctypes.c_size_t(n), None))
    _lib.check(L.mlsd_event_record(ev[1], None))
    _lib.check(L.mlsd_event_sync(ev[1]))
    ms = ctypes.c_float()
    _lib.check(L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)))
    gbs = 5 * 2 * n / (ms.value * 1e-3) / 1e9
    print(f"copy bandwidth {gbs:.0f} GB/s")
    with open(os.path.join(OUT, "probe_copy_bw.txt"), "w") as f:
        f.write(f"{gbs:.1f} GB/s\n")
    assert gbs > 1000


def test_packed_fp32_vector_instructions_do_not_run_beside_an_mfma():
    """The fact attn64x2s_kernel is designed around (round 6, tools/coissue_probe.py, profiles/r6_coissue_probe.txt): with a v_mfma_f32_32x32x16_f16 in flight (32 matrix
    clocks), three v_fma_f32 / v_exp_f32 on other registers cost almost nothing on top of it, three v_pk_fma_f32 cost more than the MFMA again.  Shader clocks of the probe
    loop per {MFMA + 3 instructions} slice, one wave per SIMD; generous margins (measured 34.5 alone, 35.5 / 37.5 with fma / exp, 61.0 with the packed form)."""
    from mlimgsynth_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    src = _lib.from_numpy(rng.standard_normal(64 * 2048 * 8).astype(np.float16))
    NB, iters = 256, 4000
    clk = _lib.DeviceBuffer(NB * 8); sink = _lib.DeviceBuffer(16)

    def clocks(mf, vk, nv):
        for _ in range(2):
            _lib.check(L.mlsd_probe_coissue(_lib.vp(src.ptr), iters, NB, 256, mf, vk, nv, _lib.vp(clk.ptr), _lib.vp(sink.ptr), None), "probe")
        _lib.check(L.mlsd_device_sync())
        return float(np.median(clk.download((NB,), np.uint64))) / (iters * 8)

    alone = clocks(1, 0, 0)
    fma, exp, pk = clocks(1, 1, 3), clocks(1, 2, 3), clocks(1, 3, 3)
    print(f"clocks per slice: MFMA alone {alone:.1f}, + 3 v_fma_f32 {fma:.1f}, + 3 v_exp_f32 {exp:.1f}, + 3 v_pk_fma_f32 {pk:.1f}")
    assert 24 < alone < 48                               # 32 matrix clocks + loop overhead
    assert fma < 1.25 * alone and exp < 1.3 * alone     # in the MFMA's shadow (measured 1.03 x / 1.09 x)
    assert pk > 1.4 * alone                             # serialised against it (measured 1.77 x)
