"""Soak of the weight-streaming plan (three slabs, segments of the next evaluation uploaded ahead): many evaluations in a row with changing inputs, each compared bit for bit
with the resident plan; tiny model with 1 MiB slabs (17 segments) and the SDXL UNet with the default 512 MiB slabs (9 segments).
usage: python3 tools/soak_stream.py [evals_tiny] [evals_sdxl]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlimgsynth_amd import engine
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for model, lat, n, mib, reps in (("tinyxl", 8, 2, 1, nt), ("sdxl", 128, 2, 512, nx)):
    rng = np.random.default_rng(5)
    res = engine.Unet(model, lat, lat, n)
    st = engine.Unet(model, lat, lat, n, stream_weights_mib=mib)
    P = res.P
    bad = 0
    for rep in range(reps):
        x = rng.standard_normal((n, 4, lat, lat)).astype(np.float32) * 3
        cond = rng.standard_normal((n, 77, P.n_ctx)).astype(np.float32)
        label = rng.standard_normal((n, P.ch_adm_in)).astype(np.float32) if P.ch_adm_in else None
        sigma = rng.uniform(0.2, 12.0, n).astype(np.float32)
        a = res.run(x, cond, label, sigma); b = st.run(x, cond, label, sigma)
        bad += not np.array_equal(a.view(np.uint32), b.view(np.uint32))
    print(f"{model} latent {lat} N={n}: {st.ctx.streaming_info()[0]} segments, {reps} evaluations, {bad} differ from the resident plan", flush=True)
    res.ctx.destroy(); st.ctx.destroy()
