"""Run N evaluations of a UNet plan (profiling target for rocprofv3: kernel trace or one --pmc pass).
usage: python3 tools/unet_eval.py <model> <latent_side> <n_batch> [evals] [oplist.txt]
With a fifth argument the plan's ops are written in launch order ("<label with shape>\\t<flop>\\t<algorithmic bytes>"), for
tools/trace_join.py to join with the per-dispatch kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import engine

model, lat, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
evals = int(sys.argv[4]) if len(sys.argv) > 4 else 10
un = engine.Unet(model, lat, lat, n, flags=16 if len(sys.argv) > 5 else 0)      # MLB_F_OPSHAPES
for _ in range(evals + 1):          # first call also autotunes the tile variants
    un.ctx.compute()
un.ctx.sync()
if len(sys.argv) > 5:
    with open(sys.argv[5], "w") as f:
        for (lab, fl), nb in zip(un.ctx.op_list(), un.ctx.op_bytes()):
            f.write(f"{lab}\t{fl:.0f}\t{nb:.0f}\n")
print("done", model, lat, n, evals)
