"""N>1 plumbing on CPU: two gloo ranks run the per-batch sequence of the image-sharded job -- mlimgsynth_amd.dist.job_step, THE
function bench.py runs -- through the library's own C entry points (mlis_amd_set_cond, mlis_amd_bcast_cond,
mlis_amd_gather_results) on engines built in the dry runtime (device memory = host memory, no kernels), with the library's
communicator over a host transport (gloo) in place of RCCL.  Only the denoising itself is a stand-in (no GPU here): each image's
"latent" depends on its own seed and on the broadcast conditioning.  The gathered result must equal the single-process
expectation (image -> seed mapping independent of the number of ranks, no per-step collective)."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LAT = 8            # tiny model, 64 x 64 pixels


def fake_latents(seeds, cond_sum):
    """stand-in for 'denoise these images': depends on each image's own seed and on the shared conditioning"""
    out = []
    for s in seeds:
        g = torch.Generator().manual_seed(int(s))
        out.append((torch.randn(4, LAT, LAT, generator=g) * cond_sum).numpy())
    return np.stack(out).astype(np.float32)


def conditioning(step, n_ctx):
    cond = (np.arange(77 * n_ctx, dtype=np.float32).reshape(77, n_ctx) / 1000 + step).astype(np.float32)
    return cond, (cond * -0.5).astype(np.float32)


def worker(rank, world, port, B, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mlimgsynth_amd import _lib, engine
    from mlimgsynth_amd import dist as mdist
    L = _lib.lib()
    L.mlsd_runtime_dry(1)
    Lh = engine._proto2()
    Lh.mlis_amd_bcast_cond.argtypes = [_lib.vp, _lib.vp, ctypes.c_int]
    Lh.mlis_amd_gather_results.argtypes = [_lib.vp, _lib.vp, ctypes.c_int, _lib.vp]
    Lh.mlis_amd_cond_device.restype = _lib.vp
    Lh.mlis_amd_cond_device.argtypes = [_lib.vp, ctypes.c_int]
    g = engine.Generator("tiny", LAT * 8, LAT * 8, B, n_step=2, defer_weights=True)     # plans only: no kernel can run here
    n_ctx = g.P.n_ctx
    comm = mdist.host_comm(L, world, rank)
    gather = _lib.DeviceBuffer(world * B * 4 * LAT * LAT * 4)
    outs = []
    for step in range(steps):
        cond, uncond = conditioning(step, n_ctx)

        def read_cond():
            c = np.empty((2 * B, 77, n_ctx), np.float32)
            L.mlsd_memcpy(c.ctypes.data_as(_lib.vp), _lib.vp(Lh.mlis_amd_cond_device(g.h, 0)), ctypes.c_size_t(c.nbytes), 1, None)
            return c

        def generate():
            c = read_cond()                      # every rank must hold rank 0's conditioning by now (cond rows, then uncond rows)
            assert np.array_equal(c[0], cond) and np.array_equal(c[B], uncond), "conditioning was not broadcast"
            lat = fake_latents(mdist.image_seeds(step, world, rank, B), float(c[0].sum()))
            L.mlsd_memcpy(_lib.vp(g.latent_ptr()), lat.ctypes.data_as(_lib.vp), ctypes.c_size_t(lat.nbytes), 0, None)
            return lat

        mdist.job_step(Lh, engine.check1, g.h, comm, world, rank, lambda: g.set_cond(cond, None, uncond, None), generate,
                       _lib.vp(gather.ptr))
        outs.append(gather.download((world * B, 4, LAT, LAT), np.float32))              # every rank holds the whole batch
    t = mdist.max_over_ranks(0.1 * (rank + 1), torch.device("cpu"))
    assert abs(t - 0.1 * world) < 1e-9
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


@pytest.mark.timeout(180)
def test_two_rank_job_step_equals_single_process():
    B, steps, world = 3, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, B, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    import queue
    for _ in range(3000):                       # (fails at once when a rank dies instead of waiting out the queue timeout)
        try:
            r, v = q.get(timeout=0.05)
            res[r] = v
        except queue.Empty:
            pass
        if len(res) == world or any(p.exitcode not in (None, 0) for p in procs):
            break
    assert len(res) == world, [p.exitcode for p in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process expectation: image i of global batch `step` has seed 42 + step*world*B + i
    exp = []
    for step in range(steps):
        cond, _ = conditioning(step, 64)
        exp.append(fake_latents([42 + step * world * B + i for i in range(world * B)], float(cond.sum())))
    exp = np.stack(exp)
    for r in range(world):
        assert np.array_equal(res[r], exp), r


def test_seed_mapping_is_world_size_independent():
    from mlimgsynth_amd import dist as mdist
    B = 4
    for world in (1, 2, 4, 8):
        # weak scaling: global batch = world*B images per step, contiguous seed blocks per rank
        seeds = [s for r in range(world) for s in mdist.image_seeds(3, world, r, B)]
        assert seeds == list(range(42 + 3 * world * B, 42 + 4 * world * B))
