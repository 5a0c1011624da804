"""Repeatability / batch-slot check of whole SDXL generations (round 5: a race of the 128x160 kernel's LayerNorm ending showed up only here).
usage: python3 tools/slot_dbg.py [steps] [generations]"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mlimgsynth_amd import engine, _lib
import golden_cases as G
cond, uncond, label, unlabel = G.gen_inputs("gen_tinyxl_8_6", "sdxl")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ngen = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = engine.Generator("sdxl", 1024, 1024, 4, n_step=steps, cfg_scale=7.0, s_ancestral=1.0, weight_seed=1234)
g.set_cond(cond, label, uncond, unlabel)
Lh = engine._proto2(); Lh.mlis_amd_handoff_retries.argtypes = [_lib.vp]
outs = []
for i in range(ngen):
    a, _ = g.generate([42, 43, 44, 45], want_images=False)
    outs.append(a.copy())
    eq = [[bool(np.array_equal(outs[j][k], a[k])) for k in range(4)] for j in range(len(outs) - 1)]
    d = [float(np.abs(outs[0][k] - a[k]).max()) for k in range(4)]
    print(f"generation {i}: per-image equal to earlier generations {eq[-3:] if eq else []}; max|diff| vs generation 0 per image {d}; hand-off retries {Lh.mlis_amd_handoff_retries(g.h)}", flush=True)
bad = any(not np.array_equal(outs[0], o) for o in outs[1:])
print("FAILED" if bad else "ok", f"({ngen} generations of {steps} steps)")
sys.exit(1 if bad else 0)
