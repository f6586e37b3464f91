"""Two ranks with the REAL aligner on ONE GPU (launched by tests/test_gpu_two_ranks.py under torch.distributed.run).

The multi-GPU path of SURVEY 8(e) -- the contig loop GSAlign.cpp:483-548 sharded over ranks, and the chunk loop GSAlign.cpp:61-94
sharded for one long contig -- has never had two GPUs to run on.  What CAN be proven on one: every rank runs the real library
(a context of its own on GPU 0), the exchange goes through torch.distributed exactly as bench.py / a host would drive it, and
rank 0 ends up with the bytes a one-rank run produces:

  --mode shard   contigs dealt by shard.assign_contigs, every finished contig staged (shard.ResultStage) and gathered on rank 0
                 (shard.gather_staged); rank 0 then aligns ALL contigs itself and compares blocks / records / both string pools
                 byte for byte.
  --mode split   one contig seeded by chunk range on both ranks (gsa_seed_chunks), hits to the owner (shard.exchange_hits),
                 owner finishes (gsa_finish_contig); compared with gsa_align_contig of the whole contig.

--backend nccl puts the exchange on RCCL with device tensors (two ranks on one device: RCCL may refuse that -- exit code 77
= "not admitted here", the test skips); --backend gloo stages through host memory.
Exit code 0 and a line "TWO_RANK_OK ..." from rank 0 = identical.
"""
import argparse
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["shard", "split"], required=True)
    ap.add_argument("--backend", choices=["gloo", "nccl"], default="gloo")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from gsalign_amd import capi, hostlib, indexio, shard, synth

    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    gdev = torch.device("cuda", 0)
    try:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=gdev)
            dev = gdev
            t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()      # (a duplicate-GPU refusal shows up at the first collective)
            assert int(t[0].item()) == world
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dev = torch.device("cpu")
    except Exception as e:      # noqa: BLE001
        print(f"TWO_RANK_SKIP backend {args.backend} with two ranks on one device: {str(e)[:300]}", flush=True)
        os._exit(77)

    tmp = tempfile.mkdtemp(prefix=f"gsa_two_rank_{rank}_")
    # the same reference and queries on every rank (deterministic), an index per rank (replicated index: SURVEY 8(e))
    if args.mode == "shard":
        refs, qrys = synth.make_pair_fast(3_000_000, 7, 0.02, seed=404)
    else:
        refs, qrys = synth.make_pair_fast(6_000_000, 1, 0.02, seed=405)
    fa = os.path.join(tmp, "ref.fa"); synth.write_fasta(fa, refs)
    px = os.path.join(tmp, "ref"); hostlib.build_index(fa, px)
    idx = indexio.load_index(px)
    g = capi.Aligner(idx, device=0)
    contigs = [np.ascontiguousarray(s) for _, s in qrys]
    ok = True; detail = ""

    if args.mode == "shard":
        deal = shard.assign_contigs([c.size for c in contigs], world)
        mine = deal[rank]
        max_items = max(len(x) for x in deal)
        lib = g.lib
        stage = shard.ResultStage(dev, upload=(lambda dst, src, n: lib.gsa_device_upload(0, C.c_void_p(dst), C.c_void_p(src), n)) if dev.type == "cuda" else None)

        def cb(k, r):
            if rank != 0:
                stage.put(0, mine[k], [(C.cast(r.blocks, C.c_void_p).value or 0, 40 * r.n_blocks), (C.cast(r.recs, C.c_void_p).value or 0, 16 * r.n_frags),
                                       (C.cast(r.aln1, C.c_void_p).value or 0, r.n_aln), (C.cast(r.aln2, C.c_void_p).value or 0, r.n_aln)])
            return 0
        capi.align_many([g], [contigs[i] for i in mine], on_result=cb, in_order=True)
        got, _ = shard.gather_staged(stage.take(0), max_items, device=dev)
        if rank == 0:
            theirs = dict(shard.parse_staged(b, capi.BLOCK_DT, capi.REC_DT) for b in got)
            want_ids = sorted(i for r in range(1, world) for i in deal[r])
            if sorted(theirs) != want_ids:
                ok = False; detail = f"contigs received {sorted(theirs)} != dealt to the other ranks {want_ids}"
            n_blocks = n_bytes = 0
            for ci in want_ids if ok else []:
                r = g.align_contig_raw(contigs[ci])
                mineb = [np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n] if n else np.zeros(0, np.uint8)
                         for p, n in ((r.blocks, 40 * r.n_blocks), (r.recs, 16 * r.n_frags), (r.aln1, r.n_aln), (r.aln2, r.n_aln))]
                t = theirs[ci]
                for name, a, b in zip(("blocks", "recs", "aln1", "aln2"), mineb, (t["blocks"].view(np.uint8).reshape(-1), t["recs"].view(np.uint8).reshape(-1), t["aln1"], t["aln2"])):
                    if a.size != b.size or not np.array_equal(a, b):
                        ok = False; detail = f"contig {ci}: {name} differs ({a.size} vs {b.size} bytes)"
                n_blocks += int(r.n_blocks); n_bytes += sum(int(x.size) for x in mineb)
                if r.n_blocks == 0:
                    ok = False; detail = f"contig {ci}: no alignment (the comparison would be empty)"
            detail = detail or f"{len(want_ids)} contigs from rank 1, {n_blocks} blocks, {n_bytes} bytes identical to the one-rank result"
    else:
        q = contigs[0]
        n_chunks = (q.size + 9999) // 10000
        b, e = shard.split_chunks(n_chunks, world)[rank]
        owner = 0
        n_hits = g.seed_chunks(q, b, e)
        moved = shard.exchange_hits(g, owner, device=dev)
        if rank == owner:
            g.finish_contig()
            got = g.blocks_as_dump(with_aln=True)
            g.align_contig(q)
            want = g.blocks_as_dump(with_aln=True)
            for k, v in want.items():
                if k not in got or got[k].shape != v.shape or not np.array_equal(got[k], v):
                    ok = False; detail = f"{k} differs"
            if want["b_score"].size == 0 or moved <= 0:
                ok = False; detail = f"nothing to compare (blocks {want['b_score'].size}, hits imported {moved})"
            detail = detail or f"chunks [{b},{e}) here + {moved} hits from rank 1 -> {want['b_score'].size} blocks, {want['aln1'].size} string bytes identical to gsa_align_contig"
        else:
            assert n_hits > 0
    g.close()
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
    dist.broadcast(flag, src=0)
    if rank == 0:
        print(("TWO_RANK_OK " if ok else "TWO_RANK_DIFFERENT ") + f"mode {args.mode} backend {args.backend}: {detail}", flush=True)
    dist.barrier(); dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
