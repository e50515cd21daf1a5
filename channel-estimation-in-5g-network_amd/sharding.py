"""Frame-range sharding for multi-GPU dataset generation (one process per GPU, RCCL over xGMI).

Frames are independent: every random draw of frame i is a pure function of (seed, i), so a
contiguous split of [0, total) over the ranks produces exactly the frames a single GPU would
(reference loop: src/dataset_generator.py:112-125, which is sequential only because it shares one
global RNG).  There is no collective on the data path; the job ends with ONE gather of the small
per-frame metadata / metric records (36 + 8 bytes per frame) to rank 0.
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int):
    """Contiguous, balanced [begin, end) of rank `rank`; the union over ranks is [0, total)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(int(total), world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def frame_choices(seed: int, frame_index: np.ndarray, num_options):
    """Counter-based replacement of the reference's four ``np.random.choice`` calls per frame
    (src/dataset_generator.py:114-117): option index k of frame i = hash(seed, i, k) mod len(options_k).
    Deterministic per global frame index, hence invariant to the sharding."""
    fi = np.asarray(frame_index, dtype=np.uint64)
    out = []
    with np.errstate(over="ignore"):
        for k, n in enumerate(num_options):
            h = _splitmix64(fi * np.uint64(4) + np.uint64(k) + _splitmix64(np.asarray([seed], dtype=np.uint64)))
            out.append((h % np.uint64(n)).astype(np.int64))
    return out


def gather_records(local: "np.ndarray", frame_index: "np.ndarray" = None, dst: int = 0, all_ranks: bool = False, group=None):
    """The job's single collective: every rank's per-frame records (float32 [n_local, C]) and, separately, their GLOBAL
    frame indices as int64 (float32 is exact only below 2^24 frames), concatenated in rank order.

    Default: a ``gather`` to rank ``dst`` (the north-star's "single gather at the end"); the other ranks get
    ``(None, None)``.  ``all_ranks=True`` makes it an ``all_gather`` for jobs in which every rank needs the summary.
    Ragged shard sizes are padded to the largest one.  Works on CPU (gloo) and GPU (nccl = RCCL) process groups; without
    an initialised group it returns the local arrays."""
    import torch
    import torch.distributed as dist
    local = np.asarray(local, dtype=np.float32)
    if local.ndim == 1:
        local = local[:, None]
    ids = np.arange(len(local), dtype=np.int64) if frame_index is None else np.asarray(frame_index, dtype=np.int64)
    if len(ids) != len(local):
        raise ValueError("frame_index and records differ in length")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local, ids
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)                           # 8 bytes per rank: the shard sizes
    counts = [int(c.item()) for c in counts]
    width, cap = local.shape[1], max(counts)
    rec = torch.zeros((cap, width), dtype=torch.float32, device=dev)
    rec[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    idt = torch.zeros(cap, dtype=torch.int64, device=dev)
    idt[: len(ids)] = torch.from_numpy(np.ascontiguousarray(ids)).to(dev)
    if all_ranks:
        rparts = [torch.zeros_like(rec) for _ in range(world)]
        iparts = [torch.zeros_like(idt) for _ in range(world)]
        dist.all_gather(rparts, rec, group=group)
        dist.all_gather(iparts, idt, group=group)
    else:
        root = rank == dst
        rparts = [torch.zeros_like(rec) for _ in range(world)] if root else None
        iparts = [torch.zeros_like(idt) for _ in range(world)] if root else None
        dist.gather(rec, rparts, dst=dst, group=group)
        dist.gather(idt, iparts, dst=dst, group=group)
        if not root:
            return None, None
    return (torch.cat([p[:c] for p, c in zip(rparts, counts)]).cpu().numpy(),
            torch.cat([p[:c] for p, c in zip(iparts, counts)]).cpu().numpy())


def broadcast_from_root(t, src: int = 0, group=None):
    """Config 5's start-up exchange: the MMSE filter W (P x P complex64, 22.5 MB at P = 1 678) is prepared once
    on ``src`` and broadcast to every rank; after that the ranks never talk until the final gather.  ``t`` is the tensor
    on ``src`` and an equally shaped, equally typed buffer elsewhere; complex tensors travel as their real view (RCCL
    has no complex type).  Returns ``t``.  No-op without an initialised process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    buf = torch.view_as_real(t) if t.is_complex() else t
    if not buf.is_contiguous():
        raise ValueError("broadcast_from_root needs a contiguous tensor")
    dist.broadcast(buf, src=src, group=group)
    return t

