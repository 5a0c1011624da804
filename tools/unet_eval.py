"""Run N evaluations of a UNet plan (profiling target for rocprofv3: kernel trace or one --pmc pass).
usage: python3 tools/unet_eval.py <model> <latent_side> <n_batch> [evals]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import engine

model, lat, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
evals = int(sys.argv[4]) if len(sys.argv) > 4 else 10
un = engine.Unet(model, lat, lat, n)
for _ in range(evals + 1):          # first call also autotunes the tile variants
    un.ctx.compute()
un.ctx.sync()
print("done", model, lat, n, evals)
