"""Same-box A/B of two library builds (alternating child processes, MLSD_LIB_PATH): UNet evaluation / decode times of the bench plans.
usage: python3 tools/ab_eval.py <libA.so> <libB.so> [rounds]      (labels: A, B; AB_ENV_A / AB_ENV_B = comma-separated NAME=VALUE settings for each side)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, os, sys
sys.path.insert(0, %r)
from mlimgsynth_amd import _lib, engine
L = _lib.lib(); vp = _lib.vp
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
st = vp(); L.mlsd_stream_create(ctypes.byref(st))
def t(ctx, k):
    for _ in range(3): ctx.compute()
    ctx.sync()
    L.mlsd_event_record(ev[0], st)
    for _ in range(k): ctx.compute()
    L.mlsd_event_record(ev[1], st); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)); return ms.value / k
out = []
if os.environ.get("AB_ONLY") == "sdxl_b8":        # (one rank's share of BASELINE configs[3]: the batch-16 plan)
    u = engine.Unet("sdxl", 128, 128, 16, stream=st.value); out.append(("sdxl_b8_unet_eval_ms", t(u.ctx, 6))); u.ctx.destroy()
    print("RESULT " + " ".join(f"{k}={v:.3f}" for k, v in out)); sys.exit(0)
if os.environ.get("AB_ONLY") == "sdxl_b4":        # (the headline plan alone: a quick A/B of one setting)
    u = engine.Unet("sdxl", 128, 128, 8, stream=st.value); out.append(("sdxl_b4_unet_eval_ms", t(u.ctx, 10))); out.append(("ln_fused", u.ctx.ln_fused() if hasattr(u.ctx, "ln_fused") else -1)); u.ctx.destroy()
    print("RESULT " + " ".join(f"{k}={v:.3f}" for k, v in out)); sys.exit(0)
u = engine.Unet("sd1", 64, 64, 2, stream=st.value, flags=8); out.append(("sd15_b1_unet_eval_ms", t(u.ctx, 30))); u.ctx.destroy()
u = engine.Unet("sdxl", 128, 128, 8, stream=st.value); out.append(("sdxl_b4_unet_eval_ms", t(u.ctx, 8))); u.ctx.destroy()
u = engine.Unet("sdxl", 128, 128, 2, stream=st.value); out.append(("sdxl_b1_unet_eval_ms", t(u.ctx, 10))); u.ctx.destroy()
u = engine.Unet("sdxl", 128, 128, 4, stream=st.value); out.append(("sdxl_b2_unet_eval_ms", t(u.ctx, 10))); u.ctx.destroy()
u = engine.Unet("sd1", 64, 64, 4, stream=st.value, flags=8); out.append(("sd15_b2_unet_eval_ms", t(u.ctx, 20))); u.ctx.destroy()
d = engine.Decoder("sdxl", 128, 128, 4, stream=st.value); out.append(("sdxl_b4_vae_decode_ms", t(d.ctx, 4))); d.ctx.destroy()
d = engine.Decoder("sd1", 64, 64, 1, stream=st.value); out.append(("sd15_b1_vae_decode_ms", t(d.ctx, 8))); d.ctx.destroy()
print("RESULT " + " ".join(f"{k}={v:.3f}" for k, v in out))
''' % ROOT
libs = {"A": sys.argv[1], "B": sys.argv[2]}
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
res = {"A": [], "B": []}
for r in range(rounds):
    for lab in ("A", "B"):
        env = dict(os.environ, MLSD_LIB_PATH=os.path.abspath(libs[lab]))
        for kv in os.environ.get("AB_ENV_" + lab, "").split(","):          # e.g. AB_ENV_A=MLSD_GN_TWO_PASS=1: the same library under two settings
            if "=" in kv: env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(lab, "failed:", p.stderr[-500:]); continue
        print(f"round {r} {lab} ({libs[lab]}): {line[0][7:]}", flush=True)
        res[lab].append(dict(kv.split("=") for kv in line[0][7:].split()))
for k in (res["A"][0].keys() if res["A"] else []):
    a = min(float(x[k]) for x in res["A"]); b = min(float(x[k]) for x in res["B"])
    print(f"{k:26s} A {a:8.3f}  B {b:8.3f}  B/A {b / a:.3f}")
