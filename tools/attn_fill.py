"""Does the 1024-token self attention lose time to its 'half-empty second round'?  (VERDICT r5 item 3 asks for a work queue on that premise.)
The 32-rows-per-wave kernel runs 128-row blocks, 3 resident per CU = 768 slots.  SDXL b4: 160 (image, head) groups x 8 blocks = 1280 blocks = 1.67 rounds.  If a round
took the same time whatever the number of co-resident blocks, 1.67 rounds would cost 2.0 (17 % idle).  This sweep times the SAME kernel at group counts that give exactly
0.5, 0.83, 1.0, 1.33, 1.67, 2.0, 2.5, 3.0 rounds: per-FLOP throughput at whole rounds against fractional rounds says what a perfect balancer could recover.
usage: python3 tools/attn_fill.py [reps]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
def timeit(fn):
    for _ in range(3): fn()
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps): fn()
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / reps
rng = np.random.default_rng(0)
tq = tk = 1024; dh = 64; nb = 8
L.mlsd_attention_force_old(1)            # the 32-rows-per-wave kernel (what the SDXL plan runs at 1024 tokens)
print("# groups = images x heads; blocks = groups x 8 (128 rows each); slots = 256 CUs x 3 resident blocks")
for heads in (6, 10, 12, 16, 20, 24, 30, 36, 48):
    D = heads * dh
    q = rng.standard_normal((nb, tq, D)).astype(np.float16); k = rng.standard_normal((nb, tk, D)).astype(np.float16); v = rng.standard_normal((nb, tk, D)).astype(np.float16)
    dq, dk, dv = _lib.from_numpy(q), _lib.from_numpy(k), _lib.from_numpy(v)
    do = _lib.DeviceBuffer(nb * tq * D * 2)
    a = kernels.AttnArgs(q=dq.ptr, k=dk.ptr, v=dv.ptr, out=do.ptr, ldq=D, ldk=D, ldv=D, ldo=D, bsq=tq * D, bsk=tk * D, bsv=tk * D, bso=tq * D, n_batch=nb, n_head=heads,
                         d_head=dh, Tq=tq, Tk=tk, causal=0)
    ts = sorted(timeit(lambda: kernels.attention(a)) for _ in range(3))
    blocks = nb * heads * 8
    print(f"groups {nb*heads:4d}  blocks {blocks:5d} = {blocks/768:4.2f} rounds: {ts[0]*1e3:7.1f} us (median {ts[1]*1e3:7.1f})  {4.0*nb*heads*tq*tk*dh/ts[0]/1e9:7.1f} TFLOP/s   "
          f"{ts[0]*1e3/(blocks/768):6.1f} us per round-equivalent", flush=True)
L.mlsd_attention_force_old(0)
