"""Plans are a pure function of (shape, compiled-in tile table): two FRESH processes produce bit-identical results.

Round 1 timed candidate tiles at first use, so two processes could pick different tiles for the same GEMM and differ in the
last bits (fp32 accumulation order).  Tile selection is now a lookup in csrc/host/tune_table.inc (or a static rule on a
miss); the timing tuner only runs offline (tools/tune_all.py, MLSD_AUTOTUNE=1).  This test runs the same generation in two
new interpreters and compares raw bytes, then a third time with the table ignored to show the comparison can fail."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from mlimgsynth_amd import engine, text, _lib
model, w, h, steps = %(model)r, %(w)d, %(h)d, %(steps)d
tc = text.TextConditioner(model, w, h, seed=1234)
cond, label, ncond, nlabel = tc.encode_pair(np.array([5, 17, 300, 42, 7], np.int32), np.array([9, 9, 8], np.int32))
g = engine.Generator(model, w, h, 2, n_step=steps, cfg_scale=7.0, s_ancestral=1.0, method="euler")
g.set_cond(cond, label, ncond, nlabel)
lat, img = g.generate([42, 43])
out = {"lat": hashlib.sha256(np.ascontiguousarray(lat).tobytes()).hexdigest(),
       "img": hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(),
       "cond": hashlib.sha256(np.ascontiguousarray(cond).tobytes()).hexdigest(),
       "misses": int(_lib.lib().mlctx_tune_misses()), "finite": bool(np.isfinite(lat).all())}
g.destroy()
print("RESULT " + json.dumps(out))
"""


def run_child(model, w, h, steps, env_extra=None):
    env = dict(os.environ)
    env.pop("MLSD_AUTOTUNE", None)
    env.update(env_extra or {})
    code = CHILD % {"root": ROOT, "model": model, "w": w, "h": h, "steps": steps}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.parametrize("model,w,h,steps", [("tinyxl", 64, 64, 6), ("sdxl", 256, 256, 2)])
def test_two_fresh_processes_are_bit_identical(model, w, h, steps):
    a = run_child(model, w, h, steps)
    b = run_child(model, w, h, steps)
    assert a["finite"] and b["finite"]
    assert a == b, (a, b)


def test_autotune_env_is_the_only_way_to_time_tiles():
    """without MLSD_AUTOTUNE no tile is ever timed: a process that ignores the table still repeats itself exactly (static
    rule), and reports its lookups as misses"""
    env = {"MLSD_TUNE_IGNORE_TABLE": "1"}
    a = run_child("tinyxl", 64, 64, 4, env)
    b = run_child("tinyxl", 64, 64, 4, env)
    assert a == b and a["misses"] > 0
