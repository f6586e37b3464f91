"""Multi-GPU sharding of query contigs.

Query contigs are independent in the reference (GSAlign.cpp:483-548: all
per-contig state is cleared at :490), so the path shards by contig with the index
replicated on every GPU and NO data-path collective.  The only exchange is the
gather of finished block records to rank 0, which writes MAF/VCF in contig order:
one all_gather of counts + one padded all_gather of the 40-byte records -- a few
KB..MB per contig, far below one xGMI link (about 153 GB/s), so a direct gather
(not a ring pipeline) is the right shape.

torch.distributed is plumbing here: backend "nccl" (= RCCL) on GPUs, "gloo" in the
CPU tests.
"""
from __future__ import annotations

import numpy as np


def assign_contigs(lengths, world: int):
    """Longest-processing-time-first: contig index lists per rank, deterministic."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i); load[r] += int(lengths[i])
    for r in range(world):
        out[r].sort()
    return out


def gather_block_records(records: np.ndarray, contig_ids: np.ndarray, device=None):
    """Gather per-rank block records to every rank (rank 0 uses them), ordered by
    (contig id, original position).  records: uint8 [n, 40] (gsa_block bytes);
    contig_ids: int32 [n].  Works without an initialised process group (world 1)."""
    import torch
    import torch.distributed as dist
    n = int(records.shape[0])
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        order = np.argsort(contig_ids, kind="stable")
        return records[order], contig_ids[order]
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    mx = max(1, max(int(c.item()) for c in cnts))
    pay = torch.zeros((mx, 44), dtype=torch.uint8, device=dev)
    if n:
        both = np.concatenate([records.reshape(n, 40), contig_ids.astype("<i4").view(np.uint8).reshape(n, 4)], axis=1)
        pay[:n] = torch.from_numpy(both).to(dev)
    pays = [torch.zeros_like(pay) for _ in range(world)]
    dist.all_gather(pays, pay)
    parts = [p[: int(c.item())].cpu().numpy() for p, c in zip(pays, cnts)]
    allb = np.concatenate(parts, axis=0) if parts else np.zeros((0, 44), np.uint8)
    ids = allb[:, 40:44].copy().view("<i4").reshape(-1)
    order = np.argsort(ids, kind="stable")
    return allb[order, :40], ids[order]
