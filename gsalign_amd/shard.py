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


def split_chunks(n_chunks: int, world: int):
    """Chunk ranges [beg, end) of one contig for `world` GPUs (seed search of a single long contig):
    contiguous, sizes differing by at most one."""
    base, extra = divmod(int(n_chunks), world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < extra else 0)
        out.append((b, e)); b = e
    return out


def exchange_hits(aligner, owner: int = 0, device=None) -> int:
    """After every rank ran `aligner.seed_chunks(contig, beg, end)` on its range: send the hits to `owner`, which imports
    them (gsa_export_hits -> dist.send/recv -> gsa_import_hits).  16 + 4 bytes per hit, point to point: on one node that is
    one xGMI link per sender (RCCL), far below its bandwidth.  Returns the number of hits the owner now holds (0 elsewhere).
    `aligner` needs export_hits() -> (uint64 keys, uint32 vals) and import_hits(keys, vals)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return -1
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    keys, vals = aligner.export_hits() if rank != owner else (np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    cnt = torch.tensor([keys.size], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    total = 0
    if rank == owner:
        for r in range(world):
            n = int(cnts[r].item())
            if r == owner or n == 0:
                continue
            tk = torch.empty(n, dtype=torch.int64, device=dev); tv = torch.empty(n, dtype=torch.int32, device=dev)
            dist.recv(tk, src=r); dist.recv(tv, src=r)
            aligner.import_hits(tk.cpu().numpy().view(np.uint64), tv.cpu().numpy().view(np.uint32))
            total += n
    elif keys.size:
        dist.send(torch.from_numpy(keys.view(np.int64)).to(dev), dst=owner)
        dist.send(torch.from_numpy(vals.view(np.int32)).to(dev), dst=owner)
    return total


def pack_result(contig: int, res: dict) -> np.ndarray:
    """One finished contig (dict with 'blocks', 'frags' structured arrays and 'aln1', 'aln2' bytes) as one uint8 record:
    everything rank 0 needs to write that contig's MAF / VCF."""
    b = np.ascontiguousarray(res["blocks"]).view(np.uint8).reshape(-1); f = np.ascontiguousarray(res["frags"]).view(np.uint8).reshape(-1)
    a1 = np.ascontiguousarray(res["aln1"], np.uint8); a2 = np.ascontiguousarray(res["aln2"], np.uint8)
    hdr = np.array([contig, b.size, f.size, a1.size, a2.size], np.int64).view(np.uint8)
    return np.concatenate([hdr, b, f, a1, a2])


def unpack_results(buf: np.ndarray, block_dt, frag_dt) -> dict:
    out, p = {}, 0
    while p < buf.size:
        contig, nb, nf, n1, n2 = (int(x) for x in buf[p:p + 40].view(np.int64)); p += 40
        blocks = buf[p:p + nb].view(block_dt).copy(); p += nb
        frags = buf[p:p + nf].view(frag_dt).copy(); p += nf
        a1 = buf[p:p + n1].copy(); p += n1
        a2 = buf[p:p + n2].copy(); p += n2
        out[contig] = dict(blocks=blocks, frags=frags, aln1=a1, aln2=a2)
    return out


def gather_results(mine: dict, block_dt, frag_dt, device=None) -> dict:
    """Every rank's finished contigs ({contig index: result dict}) on rank 0, keyed by contig index -- blocks, gap records
    AND gapped strings, so rank 0 can emit MAF / VCF in contig order exactly as a one-GPU run does.  One all_gather of the
    byte counts + one padded all_gather of the packed records (a direct gather, no ring: the payload is MBs)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(mine)
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    pay = np.concatenate([pack_result(c, r) for c, r in sorted(mine.items())]) if mine else np.zeros(0, np.uint8)
    cnt = torch.tensor([pay.size], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    mx = max(1, max(int(c.item()) for c in cnts))
    t = torch.zeros(mx, dtype=torch.uint8, device=dev)
    if pay.size:
        t[:pay.size] = torch.from_numpy(pay).to(dev)
    ts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    out = {}
    for tt, c in zip(ts, cnts):
        out.update(unpack_results(tt[: int(c.item())].cpu().numpy(), block_dt, frag_dt))
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
