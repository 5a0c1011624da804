"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards by independent units (images): the reference's batch semantic is N independent
generations with seed, seed+1, ... sharing one prompt (reference generate.sh:56-59).  Image i of the
global batch runs on rank (i // batch_per_gpu) % world with its own Philox stream (seed0 + i, offset 0),
so results do not depend on the GPU count.  Exactly one exchange at each end and NO per-step
collective: a broadcast of the text conditioning (~1.3 MB) and a gather of the final latents
(256 KiB per SDXL image), both through the library's C entry points (mlis_amd_bcast_cond / mlis_amd_gather_results:
RCCL on device buffers, or a host transport for CPU tests).
"""
import torch
import torch.distributed as dist


def image_seeds(step_idx, world, rank, batch_per_gpu, seed0=42):
    """Seeds of the images this rank generates in global batch `step_idx` (contiguous block per rank)."""
    base = seed0 + (step_idx * world + rank) * batch_per_gpu
    return [base + i for i in range(batch_per_gpu)]


def job_step(Lh, check1, engine_h, comm, world, rank, encode_fn, generate_fn, gather_ptr=None):
    """The per-batch sequence of the image-sharded job.  bench.py and the 2-rank CPU test (tests/test_dist_cpu.py) both run THIS
    function, through the same C entry points: rank 0 encodes the prompt into its engine's conditioning inputs, the conditioning
    is broadcast in place (mlis_amd_bcast_cond), every rank generates its own images, the final latents are all-gathered
    (mlis_amd_gather_results) into `gather_ptr` (world x per-rank latents, rank-major)."""
    if rank == 0:
        encode_fn()
    if world > 1:
        check1(Lh.mlis_amd_bcast_cond(engine_h, comm, 0), "mlis_amd_bcast_cond")
    out = generate_fn()
    if world > 1:
        check1(Lh.mlis_amd_gather_results(engine_h, comm, 0, gather_ptr), "mlis_amd_gather_results")
        check1(Lh.mlis_amd_sync(engine_h), "mlis_amd_sync")
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


_HOST_CB = []      # keeps the ctypes callback objects of host_comm alive


def host_comm(L, world, rank):
    """The library's communicator over a HOST transport: torch.distributed (gloo) moves the bytes.  Same interface as rccl_comm:
    the engine's mlis_amd_bcast_cond / mlis_amd_gather_results run unchanged (CPU tests in the dry runtime; machines without RCCL)."""
    import ctypes
    import numpy as np
    BC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int)
    AG = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

    def view(ptr, n):
        return torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)))

    def bcast(user, buf, nbytes, root):
        try:
            dist.broadcast(view(buf, nbytes), root)
            return 0
        except Exception:
            return 1

    def all_gather(user, send, recv, nbytes):
        try:
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, view(send, nbytes).clone())
            view(recv, nbytes * world).copy_(torch.cat(parts))
            return 0
        except Exception:
            return 1

    cb = (BC(bcast), AG(all_gather))
    _HOST_CB.append(cb)
    comm = ctypes.c_void_p()
    L.mlsd_comm_host.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, BC, AG, ctypes.c_void_p]
    rc = L.mlsd_comm_host(ctypes.byref(comm), world, rank, cb[0], cb[1], None)
    assert rc == 0, L.mlsd_last_error()
    return comm
