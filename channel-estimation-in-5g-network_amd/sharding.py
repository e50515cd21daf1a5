"""Frame-range sharding for multi-GPU dataset generation (one process per GPU, RCCL over xGMI).

Frames are independent: every random draw of frame i is a pure function of (seed, i), so a
contiguous split of [0, total) over the ranks produces exactly the frames a single GPU would
(reference loop: src/dataset_generator.py:112-125, which is sequential only because it shares one
global RNG).  There is no collective on the data path; the job ends with ONE gather of the small
per-frame metadata / metric records (36 + 8 bytes per frame, one packed block per rank) to rank 0, preceded by its
16-byte header exchange (record count and capacity of every rank).
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


def gather_records(local: "np.ndarray", frame_index: "np.ndarray" = None, dst: int = 0, all_ranks: bool = False, group=None,
                   capacity: int = None, force_collective: bool = False, total: int = None):
    """The job's single payload collective: every rank's per-frame records (float32 [n_local, C]) together with their GLOBAL frame
    indices (int64: float32 is exact only below 2^24 frames), concatenated in rank order.

    Each rank sends ONE int32 block [capacity, C + 2]: the float32 records bit-cast to int32 (the transport moves bytes, it does
    no arithmetic) followed by the low and high halves of the frame index.  The block is preceded by a 16-byte header exchange
    (an all-gather of (record count, stated capacity) per rank): from it EVERY rank learns every shard size, so the payload
    shapes agree by construction and a rank holding more records than the capacity it was told, or ranks that were told
    different capacities, make every rank of the group raise the same ``ValueError`` BEFORE the payload collective -- nobody is
    left waiting in a mismatched collective and nobody walks on into the next one alone.  ``capacity`` is the largest shard
    size; jobs that shard with ``shard_range`` pass ``total`` (the job's record count; the capacity is ``ceil(total / world)`` of
    the GROUP the collective runs in) or ``capacity`` itself; without either the header's largest count is used.

    Default: a ``gather`` to rank ``dst`` -- ``dst`` is a rank OF ``group`` (the default group: the global rank) -- the
    other ranks get ``(None, None)``.  ``all_ranks=True`` makes it an ``all_gather``.  Works on CPU (gloo) and GPU
    (nccl = RCCL) process groups.  Without an initialised group, or in a group of one, the local arrays are returned
    unless ``force_collective`` asks for the collective to run anyway (the one-rank RCCL rehearsal)."""
    import torch
    import torch.distributed as dist
    local = np.asarray(local, dtype=np.float32)
    if local.ndim == 1:
        local = local[:, None]
    ids = np.arange(len(local), dtype=np.int64) if frame_index is None else np.asarray(frame_index, dtype=np.int64)
    if len(ids) != len(local):
        raise ValueError("frame_index and records differ in length")
    ready = dist.is_available() and dist.is_initialized()
    if not ready or (dist.get_world_size(group) == 1 and not force_collective):
        return local, ids
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if not (0 <= dst < world):
        raise ValueError(f"dst {dst} is not a rank of the group (size {world})")
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    n_local, width = local.shape
    if capacity is None and total is not None:
        capacity = -(-int(total) // world)
    # header: (record count, stated capacity or -1) of every rank, 16 bytes each
    head = torch.tensor([n_local, -1 if capacity is None else int(capacity)], dtype=torch.int64, device=dev)
    heads_t = [torch.zeros_like(head) for _ in range(world)]
    dist.all_gather(heads_t, head, group=group)
    heads = [tuple(int(v) for v in h.cpu().tolist()) for h in heads_t]
    counts, caps = [h[0] for h in heads], [h[1] for h in heads]
    if len(set(caps)) != 1:                                                  # the same test, the same outcome, on every rank
        raise ValueError(f"ranks disagree about the capacity: {caps}")
    capacity = max(counts) if caps[0] < 0 else caps[0]
    bad = [r for r, c in enumerate(counts) if c > capacity]
    if bad:
        raise ValueError(f"rank(s) {bad} hold more records ({[counts[r] for r in bad]}) than the stated capacity {capacity}")
    block = np.zeros((max(capacity, 1), width + 2), dtype=np.int32)
    block[:n_local, :width] = np.ascontiguousarray(local).view(np.int32)
    block[:n_local, width:] = np.ascontiguousarray(ids).view(np.int32).reshape(-1, 2)     # little endian: low, high
    send = torch.from_numpy(block).to(dev)
    if all_ranks:
        parts = [torch.zeros_like(send) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
    else:
        root = rank == dst
        parts = [torch.zeros_like(send) for _ in range(world)] if root else None
        # torch's gather takes the GLOBAL rank of the destination
        dist.gather(send, parts, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if not root:
            return None, None
    recs, idl = [], []
    for p_, c in zip(parts, counts):
        b = p_.cpu().numpy()
        recs.append(np.ascontiguousarray(b[:c, :width]).view(np.float32))
        idl.append(np.ascontiguousarray(b[:c, width:]).view(np.int64).reshape(-1))
    return np.concatenate(recs).reshape(-1, width), np.concatenate(idl)


def broadcast_from_root(t, src: int = 0, group=None, force_collective: bool = False):
    """Config 5's start-up exchange: the MMSE filter W (P x P complex64, 22.5 MB at P = 1 678) is prepared once
    on ``src`` and broadcast to every rank; after that the ranks never talk until the final gather.  ``t`` is the tensor
    on ``src`` and an equally shaped, equally typed buffer elsewhere; complex tensors travel as their real view (RCCL
    has no complex type); ``src`` is a rank of ``group``.  Returns ``t``.  No-op without an initialised process group or in a
    group of one (unless ``force_collective``: the one-rank RCCL rehearsal)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force_collective):
        return t
    buf = torch.view_as_real(t) if t.is_complex() else t
    if not buf.is_contiguous():
        raise ValueError("broadcast_from_root needs a contiguous tensor")
    dist.broadcast(buf, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return t

