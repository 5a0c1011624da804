R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
MLSD_ENGINE_TRACE=1 python3 - <<'PY' 2>&1 | grep -E "engine|total" | tail -24
import sys, time
sys.path.insert(0, '.')
import numpy as np
from mlimgsynth_amd import engine
g = engine.Generator("sdxl", 1024, 1024, 4, n_step=20, cfg_scale=7.0, s_ancestral=1.0)
P = g.P
r = np.random.default_rng(4)
cond = r.standard_normal((77, P.n_ctx)).astype(np.float32); lab = r.standard_normal(P.ch_adm_in).astype(np.float32)
g.set_cond(cond, lab, cond * 0.5, lab)
for rep in range(2):
    t0 = time.perf_counter(); g.generate([1, 2, 3, 4], want_images=False, want_latents=False); print("total %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
PY
