"""N>1 plumbing on CPU: two gloo ranks run the same broadcast / shard / gather sequence bench.py uses
(mlimgsynth_amd/dist.py) around a stand-in for the per-image work, and the gathered result must equal the
single-process result (image -> seed mapping independent of the number of ranks, no per-step collective)."""
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


def fake_image(seed, cond):
    """stand-in for 'denoise one image': depends on the image's own seed and on the shared conditioning"""
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(4, 8, 8, generator=g) * cond.sum()


def worker(rank, world, port, B, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mlimgsynth_amd import dist as mdist
    cond = torch.zeros(2, 77, 16)
    label = torch.zeros(2, 5)
    outs = []
    for step in range(steps):
        if rank == 0:
            cond.copy_(torch.arange(2 * 77 * 16, dtype=torch.float32).reshape(2, 77, 16) / 1000 + step)
            label.fill_(step + 0.5)
        mdist.broadcast_conditioning(cond, label, 0)
        assert float(label[0, 0]) == step + 0.5                      # every rank sees rank 0's conditioning
        seeds = mdist.image_seeds(step, world, rank, B)
        local = torch.stack([fake_image(s, cond) for s in seeds])
        got = mdist.gather_latents(local, 0)
        if rank == 0:
            outs.append(torch.cat(got))
    t = mdist.max_over_ranks(0.1 * (rank + 1), torch.device("cpu"))
    assert abs(t - 0.1 * world) < 1e-9
    if rank == 0:
        q.put(torch.stack(outs).numpy())
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(120)
def test_two_rank_shard_equals_single_process():
    B, steps, world = 3, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, free_port_shared(), B, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process expectation: image i of global batch `step` has seed 42 + step*world*B + i
    exp = []
    for step in range(steps):
        cond = torch.arange(2 * 77 * 16, dtype=torch.float32).reshape(2, 77, 16) / 1000 + step
        exp.append(torch.stack([fake_image(42 + step * world * B + i, cond) for i in range(world * B)]))
    assert np.array_equal(res, torch.stack(exp).numpy())


_PORT = None


def free_port_shared():
    global _PORT
    if _PORT is None:
        _PORT = free_port()
    return _PORT


def test_seed_mapping_is_world_size_independent():
    from mlimgsynth_amd import dist as mdist
    B = 4
    for world in (1, 2, 4, 8):
        # weak scaling: global batch = world*B images per step, contiguous seed blocks per rank
        seeds = [s for r in range(world) for s in mdist.image_seeds(3, world, r, B)]
        assert seeds == list(range(42 + 3 * world * B, 42 + 4 * world * B))
