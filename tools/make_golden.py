#!/usr/bin/env python3
"""Regenerates the RNG golden vectors of tests/golden/reference_kats.json from the REFERENCE's own
Philox implementation (oracle/_ref/librng_ref.so, compiled by oracle/Makefile from
/root/reference/src/ccommon/rng_philox.c).  Only runs where /root/reference is mounted.
The sigma vectors were produced by the reference's sampling.c/unet.c during the survey
and cannot be regenerated here: those sources include ggml.h, which is not available."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

p = os.path.join(ROOT, "tests", "golden", "reference_kats.json")
g = json.load(open(p))
g["rng_seed0_offset0_n12"] = [f"{v:.8f}" for v in O.ref_randn(0, 0, 12)[0]]
g["rng_seed42_offset0_n8"] = [f"{v:.8f}" for v in O.ref_randn(42, 0, 8)[0]]
json.dump(g, open(p, "w"), indent=2)
print("updated", p)
