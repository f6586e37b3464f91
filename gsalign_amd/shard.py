"""Multi-GPU sharding of query contigs.

Query contigs are independent in the reference (GSAlign.cpp:483-548: all
per-contig state is cleared at :490), so the path shards by contig with the index
replicated on every GPU and NO data-path collective.  The only exchange is the
gather of finished results to rank 0, which writes MAF/VCF in contig order:
one all_gather of counts + point-to-point sends of the packed results to rank 0 --
far below one xGMI link (about 153 GB/s), so a direct gather (not a ring pipeline)
is the right shape.

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
    them.  8 + 4 bytes per hit, point to point: on one node that is one xGMI link per sender (RCCL), far below its bandwidth.
    On GPUs the hits never touch host memory: gsa_export_hits writes them into the send tensor (device to device), RCCL moves
    it, gsa_import_hits reads the receive tensor in place (device pointers are what both calls accept).  On CPU (gloo tests,
    stub aligner) the same exchange runs through numpy.  Returns the number of hits imported (0 elsewhere)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return -1
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    on_gpu = dev.type == "cuda" and hasattr(aligner, "hit_count")
    if rank == owner:
        keys = vals = None; n_mine = 0
    elif on_gpu:
        n_mine = aligner.hit_count()
        keys = torch.empty(max(n_mine, 1), dtype=torch.int64, device=dev); vals = torch.empty(max(n_mine, 1), dtype=torch.int32, device=dev)
        if n_mine:
            aligner.export_hits(keys.data_ptr(), vals.data_ptr())       # device -> device, on the library's stream; synchronised on return
    else:
        k, v = aligner.export_hits(); n_mine = int(k.size)
        keys = torch.from_numpy(k.view(np.int64)).to(dev); vals = torch.from_numpy(v.view(np.int32)).to(dev)
    cnt = torch.tensor([n_mine], dtype=torch.int64, device=dev)
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
            if on_gpu:
                torch.cuda.current_stream(dev).synchronize()            # (the receive is complete before the library's stream reads it)
                aligner.import_hits(tk.data_ptr(), tv.data_ptr(), n)    # in place: no host copy
            else:
                aligner.import_hits(tk.cpu().numpy().view(np.uint64), tv.cpu().numpy().view(np.uint32))
            total += n
    elif n_mine:
        dist.send(keys[:n_mine], dst=owner); dist.send(vals[:n_mine], dst=owner)
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


def gather_results(mine: dict, block_dt, frag_dt, device=None, dst: int = 0) -> dict:
    """Every rank's finished contigs ({contig index: result dict}) on rank `dst`, keyed by contig index -- blocks, gap records
    AND gapped strings, so that rank can emit MAF / VCF in contig order exactly as a one-GPU run does.  One all_gather of the
    byte counts, then every other rank SENDS its packed records to `dst` (point to point, exact sizes): only `dst` needs them,
    and at full-genome size a padded all_gather would hand every rank world x the largest payload (GBs).  Other ranks get {}."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(mine)
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    pay = np.concatenate([pack_result(c, r) for c, r in sorted(mine.items())]) if mine else np.zeros(0, np.uint8)
    cnt = torch.tensor([pay.size], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    if rank != dst:
        if pay.size:
            dist.send(torch.from_numpy(pay).to(dev), dst=dst)
        return {}
    out = unpack_results(pay, block_dt, frag_dt)
    for r in range(world):
        n = int(cnts[r].item())
        if r == dst or n == 0:
            continue
        t = torch.empty(n, dtype=torch.uint8, device=dev)
        dist.recv(t, src=r)
        out.update(unpack_results(t.cpu().numpy(), block_dt, frag_dt))
    return out


class ResultStage:
    """Finished contigs of this rank, staged for the gather to rank 0 ("grouped send/recv of block records + op
    bytes to rank 0, which writes MAF/VCF in contig order").  A worker thread hands over a finished contig as host-memory pieces
    (block records, 16-byte gap/seed records, the two gapped-string pools -- library-owned pinned memory, valid during the
    on_result callback only); put() copies them behind a 40-byte header into ONE tensor on `device` -- on a GPU through
    `upload(dst_ptr, src_ptr, nbytes)` (gsa_device_upload: pinned -> device DMA, no Python-side copy), on CPU (gloo tests) by
    memmove -- so that the gather is device to device over xGMI and nothing is packed or copied in Python."""

    HDR = 40

    def __init__(self, device, upload=None):
        import threading
        self.dev, self.upload, self.steps, self.lock = device, upload, {}, threading.Lock()

    def put(self, step: int, contig: int, pieces) -> None:
        import ctypes
        import torch
        sizes = [int(n) for _, n in pieces]
        if len(sizes) != 4:
            raise ValueError("ResultStage.put: pieces = (blocks, records, string pool 1, string pool 2)")
        pad = [(n + 7) & ~7 for n in sizes]
        t = torch.empty(self.HDR + sum(pad), dtype=torch.uint8, device=self.dev)
        hdr = np.array([contig] + sizes, np.int64)
        on_gpu = self.upload is not None and self.dev.type == "cuda"
        off = self.HDR
        if on_gpu:
            self.upload(t.data_ptr(), hdr.ctypes.data, self.HDR)
        else:
            ctypes.memmove(t.data_ptr(), hdr.ctypes.data, self.HDR)
        for (addr, n), pn in zip(pieces, pad):
            if n:
                if on_gpu:
                    self.upload(t.data_ptr() + off, addr, n)
                else:
                    ctypes.memmove(t.data_ptr() + off, addr, n)
            off += pn
        with self.lock:
            self.steps.setdefault(step, []).append((contig, t))

    def take(self, step: int):
        with self.lock:
            return sorted(self.steps.pop(step, []), key=lambda x: x[0])


def parse_staged(buf: np.ndarray, block_dt, rec_dt) -> tuple:
    """(contig, result dict) from the bytes of one staged contig (views, no copies)."""
    contig, nb, nf, n1, n2 = (int(x) for x in buf[:40].view(np.int64))
    p = 40
    blocks = buf[p:p + nb].view(block_dt); p += (nb + 7) & ~7
    recs = buf[p:p + nf].view(rec_dt); p += (nf + 7) & ~7
    a1 = buf[p:p + n1]; p += (n1 + 7) & ~7
    a2 = buf[p:p + n2]
    return contig, dict(blocks=blocks, recs=recs, aln1=a1, aln2=a2)


def gather_staged(staged, max_items: int, device=None, dst: int = 0, host_pool=None):
    """One step's staged contigs of every rank OTHER THAN `dst` -> rank `dst`, as host byte arrays: one per contig, sorted by the contig
    id in each header (parse_staged).  `dst`'s own contigs are NOT in the list -- they never leave the memory its aligner returned them
    in; a writer on `dst` merges the two by contig id.  One all_gather of the per-contig byte counts (max_items slots per rank), then
    batched point-to-point sends / receives of exact sizes (device to device: RCCL over xGMI), then -- on `dst` -- one D2H copy per
    contig into `host_pool` (a pinned uint8 tensor that grows as needed; pass the returned pool back in).  Other ranks return
    ([], host_pool)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [], host_pool
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    assert len(staged) <= max_items
    cnt = torch.zeros(max_items, dtype=torch.int64)
    for i, (_, t) in enumerate(staged):
        cnt[i] = t.numel()
    cnt = cnt.to(dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    if rank != dst:
        ops = [dist.P2POp(dist.isend, t, dst) for _, t in staged]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return [], host_pool
    recv, ops = [], []
    for r in range(world):
        if r == dst:
            continue
        for n in (int(x) for x in cnts[r].tolist()):
            if n > 0:
                t = torch.empty(n, dtype=torch.uint8, device=dev)
                recv.append(t); ops.append(dist.P2POp(dist.irecv, t, r))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    total = sum(t.numel() for t in recv)
    if dev.type == "cuda":
        if host_pool is None or host_pool.numel() < total:
            host_pool = torch.empty(int(total * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
        out, off = [], 0
        for t in recv:
            h = host_pool[off:off + t.numel()]; h.copy_(t, non_blocking=True); out.append(h); off += t.numel()
        torch.cuda.current_stream(dev).synchronize()
        got = [h.numpy() for h in out]
    else:
        got = [t.numpy() for t in recv]
    got.sort(key=lambda b: int(b[:8].view(np.int64)[0]))        # contig order, whatever rank a contig came from
    return got, host_pool


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
