#!/usr/bin/env python3
"""bench.py — images/sec of the SD denoising hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input on every GPU:
  rank 0: CLIP text encode (resident towers) -> [N>1: RCCL broadcast of cond/label over xGMI]
  every rank: 20-step Euler-ancestral denoise of `batch_per_gpu` images (cfg 7 => 40 UNet evaluations
  per image, cond+uncond batched) + latent decode (KL-VAE) -> [N>1: gather of the final latents to rank 0].
Inputs are resident in HBM when the timed region starts (weights, token ids); images are independent
units, sharded over ranks with no per-step collective (weak scaling: batch_per_gpu fixed).

Usage (driver contract):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F16_TFLOPS = 2500.0      # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_HBM_GBS = 8000.0              # HBM3E (MI355X_MICROARCH.md: ~8 TB/s)

WORKLOADS = {
    # name: (model, width, height, default batch per GPU)
    "sdxl": ("sdxl", 1024, 1024, 4),      # BASELINE.json configs[2] (and [3] at 8 GPUs)
    "sd15": ("sd1", 512, 512, 1),         # BASELINE.json configs[1]
    "tiny": ("tiny", 64, 64, 2),          # plumbing check only
    "tinyxl": ("tinyxl", 64, 64, 2),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="sdxl", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-per-gpu", type=int, default=0)
    ap.add_argument("--denoise-steps", type=int, default=20)
    ap.add_argument("--cfg", type=float, default=7.0)
    ap.add_argument("--tae", action="store_true", help="decode with TAESD instead of the KL-VAE")
    ap.add_argument("--hipgraph", type=int, default=-1, help="replay each UNet evaluation as a hipGraph (default: auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-nfe", type=int, default=2, help="(kept for compatibility: the CPU baseline sample times one sampler step = 2 UNet evaluations)")
    ap.add_argument("--cpu-timeout", type=float, default=150.0, help="hard limit (s) of each CPU-baseline child process; what it measured before that is used")
    ap.add_argument("--cpu-decode-budget", type=float, default=40.0, help="seconds the CPU baseline may spend on a full-size VAE decode; above the estimate the decode is priced, and the record says so")
    ap.add_argument("--kernel-table", default="", help="write the per-kernel time table of one UNet evaluation to this file")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads (sd15, sdxl + TAESD) reported beside the headline")
    ap.add_argument("--extra-steps", type=int, default=3)
    ap.add_argument("--extras", action="store_true", help="run the secondary workloads also when the headline workload is not sdxl (tests)")
    ap.add_argument("--transport", default="rccl", choices=("rccl", "host"),
                    help="N > 1 exchange steps: rccl = device buffers over RCCL / xGMI (the product path); host = the library's communicator over a host transport "
                         "(torch.distributed gloo), so that the world > 1 branch runs where RCCL cannot (several ranks sharing one GPU)")
    return ap.parse_args()


def pmc_files(workload, B, kind):
    """Committed rocprofv3 --pmc summaries of this plan, newest round first (profiles/r<N>_<workload>_b<B>_pmc_<kind>.json)."""
    import glob
    fs = glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_b{B}_pmc_{kind}.json"))
    rnd = lambda p_: int(re.match(r"r(\d+)_", os.path.basename(p_)).group(1)) if re.match(r"r(\d+)_", os.path.basename(p_)) else -1
    return sorted(fs, key=rnd, reverse=True)


def pmc_entry(workload, B, kind, lab):
    """(entry, file) of kernel label `lab` in the NEWEST committed summary; (None, file) when that file does not know the label --
    never an older round's number for a kernel that has changed since (tests/test_profile_labels_cpu.py keeps the newest file keyed by plan labels)."""
    for pmc in pmc_files(workload, B, kind)[:1]:
        try:
            with open(pmc) as fh:
                return json.load(fh)["kernels"].get(lab), pmc
        except Exception:
            return None, pmc
    return None, None


def kernel_roofline(g, workload, B):
    """Roofline of the dominant kernel of one UNet evaluation: per-launch HIP-event timing on the engine's stream.
    Returns (roofline dict, per-label aggregate {label: [launches, ms, flop, bytes]}, per-op ms)."""
    uc = g.unet_ctx()
    ops = uc.op_list()
    nbytes = uc.op_bytes()
    ms = uc.profile_ops()
    agg = {}
    for (lab, fl), t, nb in zip(ops, ms, nbytes):
        if t == 0.0: continue                      # a norm that runs inside its producer: no launch of its own (mlctx_profile_ops)
        lab = re.sub(r",k/\d+p?>", ">", lab)       # split-K launches run the same kernel instantiation
        e = agg.setdefault(lab, [0, 0.0, 0.0, 0.0])
        e[0] += 1; e[1] += float(t); e[2] += fl; e[3] += nb
    dom = max(agg.items(), key=lambda kv: kv[1][1])
    lab, (cnt, tms, fl, nb) = dom
    tflops = fl / (tms * 1e-3) / 1e12 if tms > 0 else 0.0
    gbs = nb / (tms * 1e-3) / 1e9 if tms > 0 else 0.0
    # the bound is the roof the kernel is closer to (GEMM/attention: MFMA; norms and short split-K GEMMs: HBM)
    if tflops / PEAK_MFMA_F16_TFLOPS >= gbs / PEAK_HBM_GBS:
        roof = {"bound": "mfma", "achieved": round(tflops, 1), "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tflops / PEAK_MFMA_F16_TFLOPS, 4)}
    else:
        roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
    # HBM-side traffic per launch of that kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same plan
    # (counters cannot be read from inside the process), summarised by tools/pmc_summary.py and committed under profiles/
    traffic, traffic_src, mfma_busy = None, None, None
    for kind in ("traffic", "mfma"):
        k, pmc = pmc_entry(workload, B, kind, lab)
        if k and kind == "traffic":
            traffic, traffic_src = k["hbm_bytes_per_launch"], os.path.relpath(pmc, ROOT)
        elif k:
            mfma_busy = k["mfma_busy_frac"]      # matrix-pipe utilisation from a SQ counter pass (tools/pmc_mfma_summary.py)
    roof = {**roof, "kernel": lab, "launches_per_eval": cnt, "avg_launch_us": round(tms / cnt * 1e3, 2),
            "algorithmic_gb_per_eval": round(nb / 1e9, 3), "algorithmic_tflop_per_eval": round(fl / 1e12, 3),
            "algorithmic_bytes_per_launch": round(nb / cnt), "traffic": traffic, "traffic_source": traffic_src, "mfma_busy_frac_pmc": mfma_busy,
            "share_of_eval_time": round(tms / float(ms.sum()), 3)}
    return roof, agg, ms


def sustained_mfma_probe(_lib, iters=20000, launches=8):
    """What this GPU's matrix pipes sustain with NOTHING else running (mlsd_probe_mfma_rate: 256 blocks x 8 waves of independent v_mfma_f32_16x16x32_f16 on random
    register operands, no memory traffic): the part lowers its clock under matrix load, so this -- not the 2.5 PFLOP/s quoted at 2.4 GHz -- is the ceiling a GEMM loop on
    this box can approach.  ~25 ms of GPU time, outside every timed region.  Reported beside `roofline.peak`, never instead of it."""
    import numpy as np
    L = _lib.lib(); vp = _lib.vp
    src = _lib.from_numpy(np.random.default_rng(0).standard_normal(64 * 2048 * 8).astype(np.float16))
    clk, sink = _lib.DeviceBuffer(256 * 8), _lib.DeviceBuffer(16)
    ev = [vp(), vp()]
    for e in ev: L.mlsd_event_create(ctypes.byref(e))
    for _ in range(2): _lib.check(L.mlsd_probe_mfma_rate(vp(src.ptr), iters, 256, vp(clk.ptr), vp(sink.ptr), None), "probe")
    L.mlsd_device_sync()
    L.mlsd_event_record(ev[0], None)
    for _ in range(launches): _lib.check(L.mlsd_probe_mfma_rate(vp(src.ptr), iters, 256, vp(clk.ptr), vp(sink.ptr), None), "probe")
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    for e in ev: L.mlsd_event_destroy(e)
    return 256 * 8 * iters * 8 * 16384.0 * launches / (ms.value * 1e-3) / 1e12


def eval_mfma(agg):
    """TIME-WEIGHTED matrix-pipe fraction of one whole UNet evaluation (sum of algorithmic FLOP of every launch / sum of launch
    durations / dense fp16 peak) and the same per kernel family, so that no bucket hides behind the best label."""
    tot_ms = sum(v[1] for v in agg.values())
    tot_fl = sum(v[2] for v in agg.values())
    fam = {}
    for lab, (c, t, fl, nb) in agg.items():
        key = ("gemm " + lab[lab.index("<") + 1:lab.index(",")] if lab.startswith("gemm<") else lab.split(" ")[0])
        e = fam.setdefault(key, [0.0, 0.0])
        e[0] += t; e[1] += fl
    top = sorted(fam.items(), key=lambda kv: -kv[1][0])[:6]
    return {"sum_launch_ms": round(tot_ms, 3), "tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
            "frac_of_mfma_peak": round(tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_MFMA_F16_TFLOPS, 4),
            "by_family": {k: {"share": round(t / tot_ms, 3), "tflops": round(fl / (t * 1e-3) / 1e12, 1)} for k, (t, fl) in top}}


def run_extra(engine, text, Lh, _lib, workload, tae, steps, cfg, denoise_steps, batch=0, split=0):
    """A secondary workload of the BASELINE metric on the same GPU (after the headline's timed region): `steps` timed steps
    (text encode + denoise + decode) after one warm-up; returns {value, ms_per_step, unet_eval_ms, roofline, ...}."""
    import numpy as np
    import torch
    model, width, height, B = WORKLOADS[workload]
    B = batch or B
    g = engine.Generator(model, width, height, B, n_step=denoise_steps, cfg_scale=cfg, s_ancestral=1.0, use_tae=tae,
                         use_hipgraph=(workload == "sd15" and not split), weight_seed=1234, unet_split=split)
    tc = text.TextConditioner(model, width, height, seed=1234)
    prompt = np.random.default_rng(7).integers(0, 49405, 8).astype(np.int32)
    pp = prompt.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))

    def step(i):
        engine.check1(Lh.mlis_amd_textcond_apply(tc.h, g.h, pp, prompt.size, None, 0), "mlis_amd_textcond_apply")
        g.generate([42 + i * B + b for b in range(B)], want_latents=False, want_images=False)

    step(steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    um = 0.0
    for i in range(steps):
        step(i)
        um += g.last_unet_ms()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    info = g.info()
    flop_per_img = (info["unet_flops"] / B) * denoise_steps + info["decode_flops"] / B
    value = steps * B / el
    roof, agg, _ = kernel_roofline(g, workload, B)
    res = {"config": f"{workload}-{width}x{height}-euler_a-{denoise_steps}-cfg{cfg:g}-b{B}-{'tae' if tae else 'vae'}",
           "value": round(value, 4), "unit": "images/s", "steps": steps, "ms_per_step": round(el / steps * 1e3, 2),
           "unet_eval_ms": round(um / (steps * denoise_steps), 3), "tflop_per_image": round(flop_per_img / 1e12, 3),
           "job_tflops": round(value * flop_per_img / 1e12, 1),
           "job_frac_of_mfma_peak": round(value * flop_per_img / 1e12 / PEAK_MFMA_F16_TFLOPS, 4),
           "roofline": roof, "unet_eval_mfma": eval_mfma(agg), "tile_table_misses": g.unet_ctx().tune_misses()}
    si = g.unet_ctx().streaming_info()
    if si:      # BASELINE configs[4]: the reference's --unet-split (weights streamed from pinned host memory through three device slabs, every evaluation)
        nseg, per_eval, slab, host = si
        evals_per_s = (steps * denoise_steps) / (um / 1e3) if um > 0 else 0.0
        res["weight_streaming"] = {"segments": nseg, "h2d_copies_per_eval": g.unet_ctx().streaming_copies(), "slab_mib": slab >> 20, "host_master_mib": host >> 20, "streamed_mib_per_eval": per_eval >> 20,
                                   "h2d_gb_per_s_sustained": round(per_eval * evals_per_s / 1e9, 1),
                                   "unet_params_on_device_mib": int(g.unet_ctx().info().mem_params) >> 20}
    aux = {"plist": g.unet_ctx().param_list(), "unet_flops_b1": info["unet_flops"] / (2 * B if cfg > 1 else B), "decode_flops_b1": info["decode_flops"] / B, "flop_per_img": flop_per_img}
    g.destroy()
    return res, aux


def cpu_sample(model, width, height, cfg, denoise_steps, threads, nfe, plist):
    """One bounded sample of the oracle (CPU restatement of the reference path): `nfe` batch-1 UNet evaluations of `model`,
    weights pre-synthesised.  Returns seconds per evaluation."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    O.L().orc_set_threads(threads)
    U = O.unet_params(model)
    OP = O.Params(1234)
    lw, lh = width // 8, height // 8
    rng = np.random.default_rng(7)
    cond = rng.standard_normal((77, U.n_ctx)).astype(np.float32)
    lab_ = rng.standard_normal(max(U.ch_adm_in, 1)).astype(np.float32)
    ot = lambda x: O.to_ot(x)
    lat = np.empty((4, lh, lw), np.float32)
    tu = ctypes.c_double()
    # synthesise the oracle's weights up front (same (seed, name, shape) rule as the engine) so that the
    # timed sample contains UNet arithmetic only
    for key, typ, ne in plist:
        OP.get(key, typ == 1, [d for d in ne[::-1]])
    n = O.L().orc_generate_latent(OP.h, b"unet", U, lw, lh, ot(cond[None, None]), ot(lab_[None, None, None]) if U.ch_adm_in else None,
                                  ot(cond[None, None]), ot(lab_[None, None, None]) if U.ch_adm_in else None,
                                  cfg, denoise_steps, 1.0, 42, nfe, O.fptr(lat), ctypes.byref(tu))
    return tu.value / max(n, 1), n


def host_threads(cap=64):
    """OpenMP threads for the CPU baseline: the CPUs this process may really use -- scheduler affinity and cgroup quota, not os.cpu_count() -- capped.  (Round 5, on the GPU
    box's 2 x 64-core / 256-thread host: the oracle's SGEMM ran at 7.4 TFLOP/s on 64 threads, 1.3 on 128 and 0.06 on 256: threads beyond the cores the box grants spin in
    OpenMP barriers.)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                       # cgroup v2
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())      # cgroup v1
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, cap))


def cpu_child(model, width, height, cfg, denoise_steps, threads, decode_budget_s, decode_flops_1, wtype="f16"):
    """CHILD PROCESS of the CPU-baseline leg (python bench.py --cpu-child ...): never touches the GPU (the plan that names the weights is built in the dry runtime).
    BASELINE.md section 3 as a bounded sample: the text towers of one prompt pair, ONE sampler step = 2 batch-1 UNet evaluations (cond + uncond) and ONE
    full-size VAE decode of the oracle (CPU restatement of the reference path), each timed and printed as soon as it is known (the parent uses what arrived before its
    timeout):   s_img = towers + (evaluations per image) x evaluation + decode."""
    import time
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from mlimgsynth_amd import _lib, engine
    _lib.lib().mlsd_runtime_dry(1)
    Lo = O.L()
    Lo.orc_set_threads(threads)
    # BASELINE configs[0] is SD1.5 **fp32**: with an fp32 checkpoint the reference keeps fp32 linear weights and ggml multiplies fp32 x fp32
    # (src/mlimgsynth.c:1235-1236, src/mlblock_nn.c:20-22) -- no F16 rounding passes over the activations in the timed sample
    Lo.orc_set_linear_wtype(0 if wtype == "f32" else 1)        # ORC_F32 = 0, ORC_F16 = 1 (oracle/oracle.h)
    U = O.unet_params(model)
    V = O.vae_params("sdxl" if model == "sdxl" else "sd1")
    OP = O.Params(1234)
    lw, lh = width // 8, height // 8
    rng = np.random.default_rng(7)
    un = engine.Unet(model, lw, lh, 2, synth=False)
    unet_flops_1 = un.ctx.info().flops / 2
    plist = un.ctx.param_list()
    emit = lambda **kw: print("CPU " + json.dumps(kw), flush=True)
    emit(threads=threads, unet_flops=unet_flops_1, decode_flops=decode_flops_1, linear_wtype=wtype)
    # ---- weights first (same (seed, name, shape) rule as the engine): the UNet's from the plan's parameter list, the VAE's / towers' by one run at 8 x 8 / as is
    for key, typ, ne in plist:
        OP.get(key, typ == 1, [d for d in ne[::-1]])
    Lo.ot_free(Lo.orc_vae_decode(OP.h, b"vae", V, O.to_ot(rng.standard_normal((1, 4, 8, 8)).astype(np.float32))))
    # SD1.5: CLIP-L, last layer + final norm; SDXL: CLIP-L and bigG at clip-skip 2 without the norm, + bigG's pooled feature (src/mlimgsynth.c:1501-1563)
    towers = [("vit_l", b"clip", 2 if model == "sdxl" else 1, model != "sdxl", False)] + ([("vit_bigg", b"clip2", 2, False, False), ("vit_bigg", b"clip2", -1, True, True)] if model == "sdxl" else [])

    def run_towers():
        for name, prefix, skip, norm, feat in towers:
            K = O.clip_params(name)
            full = np.full(K.n_token, K.tok_pad, np.int32); full[0] = K.tok_start
            full[1:10] = rng.integers(0, K.n_vocab - 3, 9); full[10] = K.tok_end
            ptr = full.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
            Lo.ot_free(Lo.orc_clip_text_encode(OP.h, prefix, K, ptr, skip, int(norm), int(feat), 10 if feat else 0))
    run_towers()                                   # (synthesises the towers' weights)
    t0 = time.perf_counter(); run_towers(); run_towers(); clip_s = time.perf_counter() - t0        # prompt + negative prompt
    emit(clip_s=clip_s)
    # ---- one sampler step: cond + uncond evaluation
    cond = rng.standard_normal((77, U.n_ctx)).astype(np.float32)
    lab_ = rng.standard_normal(max(U.ch_adm_in, 1)).astype(np.float32)
    ot = lambda x: O.to_ot(x)
    lat = np.empty((4, lh, lw), np.float32)
    tu = ctypes.c_double()
    n = Lo.orc_generate_latent(OP.h, b"unet", U, lw, lh, ot(cond[None, None]), ot(lab_[None, None, None]) if U.ch_adm_in else None,
                               ot(cond[None, None]), ot(lab_[None, None, None]) if U.ch_adm_in else None,
                               cfg, denoise_steps, 1.0, 42, 2, O.fptr(lat), ctypes.byref(tu))
    unet_eval_s = tu.value / max(n, 1)
    emit(unet_eval_s=unet_eval_s, unet_evals_timed=int(n))
    est = decode_flops_1 / (unet_flops_1 / unet_eval_s) * 2.0
    if est > decode_budget_s:
        emit(decode_s=est / 2.0, decode=f"priced at the UNet's measured FLOP rate (estimate {est:.0f} s over the {decode_budget_s:.0f} s budget)")
        return
    z = rng.standard_normal((1, 4, lh, lw)).astype(np.float32)
    t0 = time.perf_counter()
    Lo.ot_free(Lo.orc_vae_decode(OP.h, b"vae", V, O.to_ot(z)))
    emit(decode_s=time.perf_counter() - t0, decode="measured")


def cpu_record(model, width, height, cfg, denoise_steps, threads, decode_budget_s, timeout_s, flop_img, decode_flops_1, name, wtype="f16"):
    nfe = denoise_steps * (2 if cfg > 1 else 1)          # UNet evaluations per image (mlis_denoise_dxdt, src/mlimgsynth.c:1565-1587: cond + uncond)
    """Runs cpu_child in its own process (its OpenMP settings and a hard timeout never touch the product process) and composes the record from what it printed."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_WAIT_POLICY="PASSIVE", OMP_PROC_BIND="false", OMP_DYNAMIC="false")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-child", model, str(width), str(height), str(cfg), str(denoise_steps), str(threads), str(decode_budget_s), str(decode_flops_1), wtype]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    timed_out = False
    try:
        out, err = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        p.kill(); out, err = p.communicate(); timed_out = True
    rec = {}
    for line in out.splitlines():
        if line.startswith("CPU "):
            rec.update(json.loads(line[4:]))
    if "unet_eval_s" not in rec:
        raise RuntimeError(("timed out after %d s" % timeout_s if timed_out else "child failed") + ": " + (err.strip().splitlines()[-1] if err.strip() else "no output"))
    if "clip_s" not in rec:
        rec["clip_s"] = 0.0
    if "decode_s" not in rec:       # the decode did not finish inside the timeout: priced like the UNet
        rec["decode_s"] = rec["decode_flops"] / (rec["unet_flops"] / rec["unet_eval_s"]); rec["decode"] = f"priced at the UNet's measured FLOP rate (the measurement did not finish in {timeout_s} s)"
    s_img = rec["clip_s"] + nfe * rec["unet_eval_s"] + rec["decode_s"]
    return {"value": round(1.0 / s_img, 6), "unit": "images/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "s_per_unet_eval": round(rec["unet_eval_s"], 3), "s_text_towers": round(rec["clip_s"], 3), "s_vae_decode": round(rec["decode_s"], 3),
            "linear_weights": rec.get("linear_wtype", wtype), "decode": rec["decode"], "unet_gflops": round(rec["unet_flops"] / rec["unet_eval_s"] / 1e9, 1), "job_gflops": round(flop_img / s_img / 1e9, 1),
            "sample": f"{name}: text towers of a prompt pair + {rec['unet_evals_timed']} of {nfe} batch-1 UNet evaluations (one sampler step) + one VAE decode "
                      f"({rec['decode']}); linear weights {rec.get('linear_wtype', wtype)}; oracle/ = fp32 CPU restatement of the reference path, AVX-512 / AVX2 SGEMM, OpenMP {threads} threads of {os.cpu_count()} host CPUs "
                      f"(those the box grants -- cgroup quota / affinity -- capped at 64); s per image = towers + {nfe} x evaluation + decode"}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-child":
        m, w, h, cfg, ds, th, bud, dfl = sys.argv[2:10]
        return cpu_child(m, int(w), int(h), float(cfg), int(ds), int(th), float(bud), float(dfl), sys.argv[10] if len(sys.argv) > 10 else "f16")
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("multi-GPU runs are launched with torch.distributed.run (one rank per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if a.transport == "host":
        local_rank %= torch.cuda.device_count()        # ranks may share a GPU on the host transport (1-GPU boxes: both on cuda:0)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.transport == "rccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    try:
        return run(a, world, rank, local_rank)
    except BaseException:
        # a rank that fails leaves the group and exits non-zero (never re-exec: the process has initialised the GPU); the launcher ends the other ranks
        import traceback
        traceback.print_exc()
        if world > 1 and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:
                pass
        sys.exit(1)


def run(a, world, rank, local_rank):
    import numpy as np
    import torch
    import torch.distributed as dist

    from mlimgsynth_amd import _lib, engine, text
    from mlimgsynth_amd import dist as mdist
    L = _lib.lib()

    model, width, height, bdef = WORKLOADS[a.workload]
    B = a.batch_per_gpu or bdef
    hipgraph = a.hipgraph if a.hipgraph >= 0 else (1 if a.workload == "sd15" else 0)
    dev = torch.device("cuda", local_rank)

    # ---- setup (untimed): engine, weights (synthetic, seed 1234), resident text towers on rank 0
    t_setup = time.time()
    g = engine.Generator(model, width, height, B, n_step=a.denoise_steps, cfg_scale=a.cfg, s_ancestral=1.0,
                         use_tae=a.tae, use_hipgraph=bool(hipgraph), weight_seed=1234)
    P = g.P
    tc = text.TextConditioner(model, width, height, seed=1234) if rank == 0 else None
    prompt = np.random.default_rng(7).integers(0, 49405 if model in ("sd1", "sdxl") else 900, 8).astype(np.int32)
    pp = prompt.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    Lh = engine._proto2()
    Lh.mlis_amd_textcond_apply.argtypes = [_lib.vp, _lib.vp, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
    Lh.mlis_amd_bcast_cond.argtypes = [_lib.vp, _lib.vp, ctypes.c_int]
    Lh.mlis_amd_gather_results.argtypes = [_lib.vp, _lib.vp, ctypes.c_int, _lib.vp]
    comm, d_gather = None, None
    if world > 1:
        # the library owns the data-path collectives (RCCL over xGMI through its C entry points); torch.distributed only carries
        # the 128-byte unique id to the other ranks and provides the barrier / max-over-ranks of the timing contract
        comm = mdist.rccl_comm(L, world, rank, dev) if a.transport == "rccl" else mdist.host_comm(L, world, rank)
        d_gather = _lib.DeviceBuffer(world * B * 4 * (height // 8) * (width // 8) * 4)
    info = g.info()
    t_setup = time.time() - t_setup

    def one_step(idx, want_images=False):
        # rank 0: encode prompt + (empty) negative prompt straight into the plan's inputs (resident CLIP towers); N>1: ~1.3 MB
        # broadcast once per batch, device to device; every rank: its own images (independent Philox stream per image, seed 42 +
        # global image index: results do not depend on the GPU count); N>1: all-gather of the final latents (256 KiB per image)
        return mdist.job_step(
            Lh, engine.check1, g.h, comm, world, rank,
            lambda: engine.check1(Lh.mlis_amd_textcond_apply(tc.h, g.h, pp, prompt.size, None, 0), "mlis_amd_textcond_apply"),
            lambda: g.generate(mdist.image_seeds(idx, world, rank, B), want_latents=False, want_images=want_images),   # syncs its stream
            _lib.vp(d_gather.ptr) if d_gather else None)

    tdev = dev if a.transport == "rccl" else torch.device("cpu")      # where torch.distributed's own tensors live (gloo: host)
    comm_ranks, comm_kind = ctypes.c_int(0), ctypes.c_int(-1)
    if world > 1:
        assert L.mlsd_comm_count(comm, ctypes.byref(comm_ranks), ctypes.byref(comm_kind)) == 0, _lib.last_error()
        assert comm_ranks.value == world, f"the communicator reports {comm_ranks.value} ranks, the launcher {world}"

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_step(a.steps + i)           # non-negative image indices (negative ones wrapped in the uint64 seeds)
    fence()
    t0 = time.perf_counter()
    unet_ms = 0.0
    for i in range(a.steps):
        one_step(i)
        unet_ms += g.last_unet_ms()
    fence()
    el = time.perf_counter() - t0
    el = mdist.max_over_ranks(el, tdev)
    # PCIe-inclusive variant (the reference's mlis_generate ends with host pixels): one more step that also copies the fp32
    # images of this rank to host memory; reported beside `value`, never as `value`
    fence()
    t1 = time.perf_counter()
    one_step(a.steps + a.warmup, want_images=True)
    fence()
    el_host = mdist.max_over_ranks(time.perf_counter() - t1, tdev)

    if rank != 0:
        if world > 1:
            dist.barrier()
            L.mlsd_rccl_destroy(comm)
            dist.destroy_process_group()
        return

    images = a.steps * B * world
    value = images / el
    nfe_per_img = 2 * a.denoise_steps if a.cfg > 1 else a.denoise_steps
    flop_per_img = (info["unet_flops"] / B) * a.denoise_steps + info["decode_flops"] / B     # unet_flops is per batched evaluation
    out = {
        "metric": f"images/sec ({a.denoise_steps}-step Euler-a, cfg {a.cfg:g}) {'SDXL 1024x1024' if a.workload == 'sdxl' else a.workload}",
        "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(el / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{a.workload}-{width}x{height}-euler_a-{a.denoise_steps}-cfg{a.cfg:g}-b{B}-{'tae' if a.tae else 'vae'}",
                   "batch_per_gpu": B, "global_batch": B * world, "unet_evals_per_image": nfe_per_img,
                   "tflop_per_image": round(flop_per_img / 1e12, 3), "hipgraph": bool(hipgraph),
                   "parallelism": (f"image-sharded x{world}, " + ("RCCL bcast cond + gather latents" if a.transport == "rccl" else "HOST transport (gloo) bcast cond + gather latents: not the product path")) if world > 1 else "single GPU",
                   "weights": "synthetic seed 1234", "setup_s": round(t_setup, 1)},
        "job_tflops": round(value * flop_per_img / 1e12, 1),
        "job_frac_of_mfma_peak": round(value * flop_per_img / 1e12 / (world * PEAK_MFMA_F16_TFLOPS), 4),
        "unet_eval_ms": round(unet_ms / (a.steps * a.denoise_steps), 3),
        "images_per_s_with_d2h_of_fp32_images": round(B * world / el_host, 4),
    }
    if world > 1:
        out["transport"] = a.transport
        out["rccl_ranks" if comm_kind.value == 0 else "host_transport_ranks"] = comm_ranks.value      # the communicator's world as the transport reports it (ncclCommCount)
        out["gpus_visible"] = torch.cuda.device_count()

    roof, agg, ms = kernel_roofline(g, a.workload, B)
    if rank == 0 and roof.get("bound") == "mfma":
        try:       # the power-limited ceiling of THIS box, measured: pure-MFMA loop on random operands (see sustained_mfma_probe)
            sus = sustained_mfma_probe(_lib)
            roof["sustained_mfma_tflops_measured"] = round(sus, 1)
            roof["frac_of_sustained_measured"] = round(roof["achieved"] / sus, 4)
        except Exception as e:
            roof["sustained_mfma_tflops_measured"] = None; roof["sustained_probe_error"] = str(e)
    out["roofline"] = roof
    out["unet_eval_mfma"] = eval_mfma(agg)
    out["tile_table_misses"] = g.unet_ctx().tune_misses()     # GEMM shapes of this plan not in the compiled-in tile table (0 on the bench plans)
    if a.kernel_table:
        with open(a.kernel_table, "w") as f:
            f.write(f"# one UNet evaluation, {a.workload} batch {B} (N={2 * B if a.cfg > 1 else B}); per-launch HIP events\n")
            f.write(f"# {'kernel':44s} {'launches':>8s} {'total_ms':>10s} {'TFLOP':>9s} {'TFLOP/s':>9s} {'alg_GB':>9s} {'GB/s':>9s}\n")
            for k, (c, t, fl_, nb_) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{k:46s} {c:8d} {t:10.3f} {fl_ / 1e12:9.3f} {fl_ / max(t, 1e-9) / 1e9:9.1f} {nb_ / 1e9:9.3f} {nb_ / max(t, 1e-9) / 1e6:9.1f}\n")

    # ---- secondary workloads of the BASELINE metric (N=1 only, after the headline's timed region; the headline `value` is
    # untouched): SD1.5 512x512 batch 1 (configs[1]) and SDXL with the TAESD decoder (configs[4] without weight streaming)
    plist = g.unet_ctx().param_list()
    extras = world == 1 and not a.no_extras and ((a.workload == "sdxl" and not a.tae) or a.extras)
    aux15 = None
    if extras:
        g.destroy()
        # sdxl_b8 = ONE RANK's share of BASELINE configs[3] (8 GPUs x 8 images: the batch-16 UNet plan): the per-GPU number the
        # 8-GPU job multiplies, measured on this GPU with its own roofline
        for key, wl, tae_, b_, st_ in (("sd15", "sd15", False, 0, a.extra_steps), ("sdxl_tae", "sdxl", True, 0, a.extra_steps),
                                       ("sdxl_b8", "sdxl", False, 8, min(a.extra_steps, 2)),
                                       # configs[4] as the reference runs it: TAESD decode + --unet-split (UNet weights streamed, not resident)
                                       ("sdxl_tae_split", "sdxl", True, 0, min(a.extra_steps, 2))):
            try:
                out[key], aux = run_extra(engine, text, Lh, _lib, wl, tae_, st_, a.cfg, a.denoise_steps, batch=b_, split=1 if key == "sdxl_tae_split" else 0)
                if key == "sd15":
                    aux15 = aux
            except Exception as e:
                out[key] = {"value": None, "error": str(e)}

    # ---- CPU baseline (rank 0, N=1 only): the oracle = CPU restatement of the reference path, bounded samples of the WHOLE path (towers, sampler step, decode), each in
    # a child process with a hard timeout
    if world == 1 and not a.no_cpu_baseline:
        threads = a.cpu_threads or host_threads()
        try:
            out["cpu_baseline"] = cpu_record(model, width, height, a.cfg, a.denoise_steps, threads, a.cpu_decode_budget, a.cpu_timeout, flop_per_img, info["decode_flops"] / B, f"one {a.workload} {width}x{height} image")
        except Exception as e:  # the baseline is a report, never the product path
            out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        # BASELINE.md section 3 / configs[0] name SD1.5 fp32 512x512 on the CPU: the same sample of that path
        if aux15:
            try:
                out["cpu_baseline"]["sd15"] = cpu_record("sd1", 512, 512, a.cfg, a.denoise_steps, threads, a.cpu_decode_budget, a.cpu_timeout, aux15["flop_per_img"], aux15["decode_flops_b1"], "one SD1.5 fp32 512x512 image (BASELINE configs[0])", wtype="f32")
            except Exception as e:
                out["cpu_baseline"]["sd15"] = {"value": None, "sample": f"failed: {e}"}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        L.mlsd_rccl_destroy(comm)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
