"""Multi-GPU sharding of the pair list: one process per GPU, a static partition of the image
pairs, and a gather of the per-rank CSR match lists to rank 0.

Pairs are independent (the reference's match() and RANSAC() read only their own two images,
submodules/scene.cpp:60,261), so the data path has no collective; the only exchange is the final
gather of results.  Partition and gather are the library's (sfm_shard_*, sfm_comm_*, sfm_match_gather
in include/sfm_amd.h: RCCL over xGMI, counts by all-gather, lists point to point into a rank-major
buffer on the root); `partition_pairs` and `library_gather` below call them.  `gather_csr` is the same
exchange and the same layout written with torch.distributed, for process groups without a GPU (the CPU
tests run it on "gloo") and as a stand-by on "nccl".
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def partition_pairs(n_pairs: int, rank: int, world: int, block: int = 64) -> np.ndarray:
    """Indices (ascending) of the pairs rank owns: blocks of `block` consecutive pairs dealt
    round-robin, so ranks differ by at most one block and consecutive pairs (which share an
    image in an exhaustive list) stay on one GPU."""
    if block == 64:  # the library's partition (sfm_shard_indices)
        from . import capi
        return capi.shard_indices(n_pairs, rank, world)
    if world <= 1:
        return np.arange(n_pairs, dtype=np.int64)
    idx = np.arange(n_pairs, dtype=np.int64)
    return idx[(idx // block) % world == rank]


def library_comm(ctx, rank: int, world: int):
    """The library's communicator for this rank, its id handed round through torch.distributed (any
    initialised process group will do; world 1 needs none)."""
    from . import capi
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if world > 1:
            raise RuntimeError("library_comm: more than one rank needs an initialised torch.distributed process group")
        return capi.Comm.create(ctx, 0, 1, capi.Comm.unique_id())
    box = [capi.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return capi.Comm.create(ctx, rank, world, box[0])


def library_gather(comm, result, n_pairs_total: int, gathered=None):
    """sfm_match_gather: this rank's result (over partition_pairs(n_pairs_total, rank, world), in that
    order) to rank 0.  Returns the reusable capi.Gathered (its lists live on rank 0's device)."""
    return comm.gather(result, n_pairs_total, 0, gathered)


def images_for_pairs(n_pairs: int) -> int:
    """Smallest n with n*(n-1)/2 >= n_pairs."""
    n = int(np.ceil((1 + np.sqrt(1 + 8.0 * n_pairs)) / 2))
    while n * (n - 1) // 2 < n_pairs:
        n += 1
    while n > 2 and (n - 1) * (n - 2) // 2 >= n_pairs:
        n -= 1
    return n


class DevicePtrView:
    """Zero-copy view of library-owned device memory for torch.as_tensor()."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def gather_csr(counts, payload, n_pairs_total: int, rank: int, world: int, block: int = 64,
               width: int = 2, device=None, ordered: bool = True) -> Optional[Tuple[np.ndarray, ...]]:
    """Gather per-rank CSR lists to rank 0.

    counts: torch int64 tensor [owned pairs] (entries per owned pair, on `device`);
    payload: torch int32 tensor [sum(counts), width] on `device`.
    Returns None on ranks other than 0.  On rank 0:
      ordered=True:  (offsets int64 numpy [n_pairs_total+1], payload tensor in global pair order);
      ordered=False: (starts int64 numpy [n_pairs_total], counts int64 numpy [n_pairs_total], payload
                     tensor laid out rank after rank): pair p owns payload[starts[p] : starts[p] + counts[p]].
                     Every peer's list is received straight into its place, nothing is copied twice.
    Uses torch.distributed send/recv (RCCL over xGMI with the nccl backend: rank 0 posts all
    receives at once, one per peer link)."""
    import torch
    import torch.distributed as dist

    if world <= 1:
        c = counts.cpu().numpy().astype(np.int64)
        off = np.zeros(n_pairs_total + 1, np.int64)
        np.cumsum(c, out=off[1:])
        return (off, payload) if ordered else (off[:-1].copy(), c, payload)

    owned = [partition_pairs(n_pairs_total, r, world, block) for r in range(world)]
    max_owned = max(len(o) for o in owned)
    # step 1: per-pair counts of every rank (fixed-size all_gather)
    mine = torch.zeros(max_owned, dtype=torch.int64, device=counts.device)
    mine[:len(counts)] = counts
    all_counts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(all_counts, mine)
    # step 2: payloads to rank 0, all peers at once, each into its region of one buffer
    # (Both sides use batched point-to-point operations: on the nccl backend those run on the group's own
    # communicator, whereas a plain send would open a two-rank communicator that rank 0, which receives in
    # a batch, never joins, unless the group was created with device_id.)
    if rank != 0:
        if payload.numel() > 0:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, payload.contiguous(), 0)]):
                req.wait()
        return None
    per_rank_counts = [all_counts[r][:len(owned[r])].cpu().numpy().astype(np.int64) for r in range(world)]
    totals = [int(c.sum()) for c in per_rank_counts]
    base = np.concatenate([[0], np.cumsum(totals)]).astype(np.int64)
    gathered = torch.empty((int(base[-1]), width), dtype=payload.dtype, device=payload.device)
    if totals[0] > 0:
        gathered[:totals[0]] = payload
    ops = []
    for r in range(1, world):
        if totals[r] > 0:
            ops.append(dist.P2POp(dist.irecv, gathered[int(base[r]):int(base[r + 1])], r))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    global_counts = np.zeros(n_pairs_total, np.int64)
    starts = np.zeros(n_pairs_total, np.int64)
    for r in range(world):
        global_counts[owned[r]] = per_rank_counts[r]
        if len(owned[r]):
            starts[owned[r]] = base[r] + np.concatenate([[0], np.cumsum(per_rank_counts[r])[:-1]])
    if not ordered:
        return starts, global_counts, gathered
    # step 3: global CSR in pair order
    off = np.zeros(n_pairs_total + 1, np.int64)
    np.cumsum(global_counts, out=off[1:])
    out = torch.empty_like(gathered)
    for r in range(world):
        if len(owned[r]) == 0:
            continue
        # consecutive owned pairs of one block are contiguous on both sides: copy block-wise
        first = np.flatnonzero(np.diff(np.concatenate([[-2], owned[r]])) != 1)
        last = np.concatenate([first[1:], [len(owned[r])]]) - 1
        for s, e in zip(first, last):
            a = int(starts[owned[r][s]])
            b = int(starts[owned[r][e]] + global_counts[owned[r][e]])
            if b > a:
                d0 = int(off[owned[r][s]])
                out[d0:d0 + (b - a)] = gathered[a:b]
    return off, out
