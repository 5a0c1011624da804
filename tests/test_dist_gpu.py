"""BASELINE configs[3] on real device memory with the ONE GPU a box has: two ranks (two processes, both on cuda:0) run the per-batch sequence of the image-sharded job --
mlimgsynth_amd.dist.job_step, THE function bench.py runs per step -- through the library's C entry points (mlis_amd_bcast_cond, mlis_amd_generate,
mlis_amd_gather_results) with the library's communicator over a HOST transport (gloo; RCCL refuses two ranks on one device, and the 8-GPU run is the driver's).  Unlike
tests/test_dist_cpu.py (dry runtime, stand-in latents) the engines here denoise for real: every rank ends with the WHOLE global batch, and it must equal, bit for bit,
what ONE process generates for the same images with the per-rank plan -- image i has its own Philox stream (seed 42 + i, reference generate.sh:56-59), the conditioning is
encoded on rank 0 only and broadcast device to device -- and, within the generation bound, what one process generates for the global batch as ONE batch (another plan: other
tiles, other fp32 summation orders): results do not depend on the number of GPUs."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

MODEL, SIDE, STEPS_DENOISE = "tinyxl", 64, 3


def conditioning(step, P):
    r = np.random.default_rng(100 + step)
    cond = r.standard_normal((77, P.n_ctx)).astype(np.float32)
    lab = r.standard_normal(P.ch_adm_in).astype(np.float32) if P.ch_adm_in else None
    return cond, lab, (cond * 0.25).astype(np.float32), lab


def worker(rank, world, port, B, steps, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mlimgsynth_amd import _lib, engine
    from mlimgsynth_amd import dist as mdist
    L = _lib.lib()
    Lh = engine._proto2()
    Lh.mlis_amd_bcast_cond.argtypes = [_lib.vp, _lib.vp, ctypes.c_int]
    Lh.mlis_amd_gather_results.argtypes = [_lib.vp, _lib.vp, ctypes.c_int, _lib.vp]
    g = engine.Generator(MODEL, SIDE, SIDE, B, n_step=STEPS_DENOISE, cfg_scale=7.0, s_ancestral=1.0)
    comm = mdist.host_comm(L, world, rank)
    lat = SIDE // 8
    gather = _lib.DeviceBuffer(world * B * 4 * lat * lat * 4)
    outs = []
    for step in range(steps):
        cond, lab, uncond, unlab = conditioning(step, g.P)
        junk = np.full_like(cond, 7.0)                       # ranks other than 0 hold junk until the broadcast
        if rank:
            g.set_cond(junk, lab * 0 if lab is not None else None, junk, lab * 0 if lab is not None else None)
        mdist.job_step(Lh, engine.check1, g.h, comm, world, rank, lambda: g.set_cond(cond, lab, uncond, unlab),
                       lambda: g.generate(mdist.image_seeds(step, world, rank, B), want_latents=False, want_images=False), _lib.vp(gather.ptr))
        outs.append(gather.download((world * B, 4, lat, lat), np.float32))
    q.put((rank, np.stack(outs)))
    dist.barrier()
    assert L.mlsd_rccl_destroy(comm) == 0
    g.destroy()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_generate_the_single_process_batch():
    import queue
    import torch.multiprocessing as mp
    B, steps, world = 2, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, B, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(10000):
        try:
            r, v = q.get(timeout=0.05)
            res[r] = v
        except queue.Empty:
            pass
        if len(res) == world or any(p.exitcode not in (None, 0) for p in procs):
            break
    assert len(res) == world, [p.exitcode for p in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # the same global batches from ONE process.  (a) with the per-rank plan (batch B), rank by rank: bit-identical -- same launches, same seeds, the conditioning each rank got by
    # broadcast; (b) as ONE batch of world x B images (another plan: other tiles and K splits, so other fp32 summation orders): within the generation bound of tests/tolerances.py
    import tolerances as T
    from mlimgsynth_amd import engine
    gb = engine.Generator(MODEL, SIDE, SIDE, B, n_step=STEPS_DENOISE, cfg_scale=7.0, s_ancestral=1.0)
    gw = engine.Generator(MODEL, SIDE, SIDE, world * B, n_step=STEPS_DENOISE, cfg_scale=7.0, s_ancestral=1.0)
    for step in range(steps):
        cond, lab, uncond, unlab = conditioning(step, gb.P)
        gb.set_cond(cond, lab, uncond, unlab); gw.set_cond(cond, lab, uncond, unlab)
        exp = np.concatenate([gb.generate([42 + (step * world + r) * B + i for i in range(B)], want_images=False)[0] for r in range(world)])
        whole, _ = gw.generate([42 + step * world * B + i for i in range(world * B)], want_images=False)
        assert np.isfinite(exp).all()
        for r in range(world):
            assert np.array_equal(res[r][step].view(np.uint32), exp.view(np.uint32)), (step, r)
        e = np.linalg.norm(whole.astype(np.float64) - exp) / np.linalg.norm(exp)
        print(f"global batch {step}: 2 ranks x {B} images vs one batch of {world * B}: rel-L2 {e:.2e}")
        assert e < T.LATENT
    gb.destroy(); gw.destroy()
