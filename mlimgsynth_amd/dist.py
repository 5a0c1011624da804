"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards by independent units (images): the reference's batch semantic is N independent
generations with seed, seed+1, ... sharing one prompt (reference generate.sh:56-59).  Image i of the
global batch runs on rank (i // batch_per_gpu) % world with its own Philox stream (seed0 + i, offset 0),
so results do not depend on the GPU count.  Exactly one exchange at each end and NO per-step
collective: a broadcast of the text conditioning (~1.3 MB) and a gather of the final latents
(256 KiB per SDXL image).
"""
import torch
import torch.distributed as dist


def image_seeds(step_idx, world, rank, batch_per_gpu, seed0=42):
    """Seeds of the images this rank generates in global batch `step_idx` (contiguous block per rank)."""
    base = seed0 + (step_idx * world + rank) * batch_per_gpu
    return [base + i for i in range(batch_per_gpu)]


def broadcast_conditioning(cond, label, src=0):
    """cond [2][77][n_ctx] (prompt, negative prompt), label [2][adm] or None; in place on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    dist.broadcast(cond, src)
    if label is not None:
        dist.broadcast(label, src)


def gather_latents(local, dst=0):
    """local [B][4][h][w] on every rank -> list of world tensors on rank dst (None elsewhere)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    world, rank = dist.get_world_size(), dist.get_rank()
    out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, out, dst=dst)
    return out


def max_over_ranks(seconds, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rccl_comm(L, world, rank, device):
    """RCCL communicator of the C library (mlsd_rccl_*): rank 0 draws the 128-byte unique id, torch.distributed carries it to the
    other ranks (the only thing it is used for on the data path), every rank joins.  Returns the opaque communicator handle."""
    import ctypes
    buf = ctypes.create_string_buffer(128)
    if rank == 0:
        rc = L.mlsd_rccl_unique_id(buf)
        assert rc == 0, L.mlsd_last_error()
    t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    if dist.get_backend() == "nccl":
        t = t.to(device)
    dist.broadcast(t, 0)
    raw = bytes(t.cpu().numpy().tobytes())
    comm = ctypes.c_void_p()
    rc = L.mlsd_rccl_init(ctypes.byref(comm), world, rank, raw)
    assert rc == 0, L.mlsd_last_error()
    return comm
