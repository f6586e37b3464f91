#!/usr/bin/env python3
"""BASELINE configs[4] on ONE GPU (run by tests/test_gpu_parity.py::test_config5_full_human_all_contigs in a process of its own): a
24-contig reference with GRCh38 chromosome lengths (3.08 Gbp, 6.2 G BWT rows: the >= 2^32-row device layout and the 64-bit suffix sorter
on their real input), query = 1 %-diverged copy of every chromosome (one of them reverse-complemented), -alen 5000.
  1. all contigs through gsa_align_many on two contexts: throughput, then EVERY contig through the result invariants;
  2. ORACLE PARITY ON THE NATIVE >= 2^32-ROW INDEX (round 5): whole contigs and pieces -- forward strand, reverse strand, reference positions
     above 2^32 -- against the real reference (oracle/_ref/libgsref.so loads the same index files; the CPU restatement when it is absent),
     every stage dump S1..S8 incl. both gapped-string pools, bit for bit;
  3. a bundle of short contigs against the human-sized index == the same contigs one by one.
Test infrastructure (it loads oracle/): lives under tests/.  GPU box only (host: ~120 GB, HBM: ~110 GB).
usage: human_scale_check.py [scale=1.0]"""
import os, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gsalign_amd import synth, hostlib, indexio, capi
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
MB = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57]
lens = [int(m * 1e6 * scale) for m in MB]
tmp = tempfile.mkdtemp(prefix="human_", dir="/tmp")
try:
    t = time.time()
    refs = []
    for i, n in enumerate(lens):
        r = synth.fast_genome(n, 41000 + i); synth.inject_repeats(r, 41000 + i); refs.append((f"chr{i + 1}", r))
    synth.write_fasta(os.path.join(tmp, "r.fa"), refs)
    print(f"reference: {len(lens)} contigs, {sum(lens)} bp, written in {time.time() - t:.0f} s", flush=True)
    t = time.time(); hostlib.build_index(os.path.join(tmp, "r.fa"), os.path.join(tmp, "r")); print(f"index built in {time.time() - t:.0f} s", flush=True)
    t = time.time(); idx = indexio.load_index(os.path.join(tmp, "r")); print(f"index loaded in {time.time() - t:.0f} s, seq_len {idx.seq_len}", flush=True)
    t = time.time(); g0 = capi.Aligner(idx, alen=5000); g1 = g0.clone(); print(f"gsa_create {time.time() - t:.0f} s", flush=True)
    # (round 6) a second, independent copy of the finished device index, device to device: what every further GPU of a node costs instead of gsa_create
    t = time.time(); g_copy = g0.clone_to_device(0); dt_copy = time.time() - t
    print(f"gsa_clone_to_device (same device, {idx.G / 1e9:.2f} Gbp index): {dt_copy:.2f} s", flush=True)
    qs = [synth.fast_mutate(r, 0.01, 51000 + i) for i, (_, r) in enumerate(refs)]
    qs[20] = synth.revcomp(qs[20])
    pinned = [g0.pinned_copy(q) for q in qs]
    total = sum(q.size for q in qs)
    # ---- round 6: EVERY contig of the genome against the real reference (GenomeComparison's loop, GSAlign.cpp:473-552, on the index files its own loader
    # reads, bwt_index.cpp:147-264): P reference processes side by side, one thread and one copy of the index (~11 GB) each, started now and collected
    # after the GPU passes below -- the seeds (stage 1) and the final result with both gapped-string pools (stage 8) of all 24 contigs
    from oracle import oracle_py as op
    PRM = dict(alen=5000)
    ref_jobs = []
    if op.have_ref():
        avail = 0.0
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                avail = int(ln.split()[1]) / 1e6
        P = int(max(1, min(16, (avail - 80.0) // 16, (os.cpu_count() or 8) // 4, len(qs))))
        from gsalign_amd import shard
        deal = shard.assign_contigs([q.size for q in qs], P)
        for k, own in enumerate(deal):
            if not own:
                continue
            qfa, npz = os.path.join(tmp, f"all_q{k}.fa"), os.path.join(tmp, f"all_{k}.npz")
            synth.write_fasta(qfa, [(f"q{ci}", qs[ci]) for ci in own])
            ref_jobs.append((own, npz, op.ref_dump_subprocess(os.path.join(tmp, "r"), qfa, npz, PRM, upto=8, stages=(1, 8), wait=False)))
        t_ref0 = time.time()
        print(f"reference side of the whole-genome comparison: {len(ref_jobs)} processes started ({avail:.0f} GB of host memory available)", flush=True)
    devq = [g0.device_copy(q) for q in qs]
    # ... and it answers like the original: one contig through the copy == through the owner, byte for byte
    ra = g0.align_contig(pinned[21]); da = capi.result_as_dump(ra, with_aln=True)
    rb = g_copy.align_contig(pinned[21]); db = capi.result_as_dump(rb, with_aln=True)
    for key in da:
        assert np.array_equal(da[key], db[key]), ("clone_to_device", key)
    g_copy.close()
    print("the copied index answers like the original (contig 21, every block, record and string byte)", flush=True)
    for rep in range(2):
        t = time.time(); capi.align_many([g0, g1], pinned); dt = time.time() - t
        print(f"pass {rep}: {len(qs)} contigs, {total} bp in {dt * 1e3:.0f} ms = {total / dt / 1e9:.2f} Gbp/s (H2D and D2H included, 2 contexts)", flush=True)
    for rep in range(2):
        t = time.time(); capi.align_many([g0, g1], devq); dt = time.time() - t
        print(f"pass {rep}, contigs resident in HBM: {total} bp in {dt * 1e3:.0f} ms = {total / dt / 1e9:.2f} Gbp/s (D2H included, 2 contexts)", flush=True)
    from test_gpu_parity import _check_result_invariants
    # every contig: result invariants (records tile their blocks, seeds are exact matches, the gapped strings spell both fragments,
    # lengths and scores are what the strings say); the oracle comparison follows below
    tot_cov = 0
    for ci in range(len(qs)):
        t = time.time(); g0.align_contig(pinned[ci]); dt = time.time() - t
        r = g0.blocks(); _check_result_invariants(idx, qs[ci], r)
        cov = int(r["blocks"]["aln_len"].sum()); tot_cov += cov
        print(f"contig {ci}: {qs[ci].size} bp in {dt * 1e3:.1f} ms, {r['blocks'].size} blocks, coverage {cov / qs[ci].size:.3f}, invariants ok", flush=True)
        # (contig 20 is reverse-complemented and cut in two by the tandem array in its middle: the reference's second redundancy pass drops one
        #  of two reverse-strand blocks of one chromosome -- SURVEY App. B #11, reproduced; tests/test_gpu_parity.py::test_long_kmer_table has
        #  the same shape against the oracle)
        assert cov > (0.45 if ci == 20 else 0.9) * qs[ci].size
    print(f"all {len(qs)} contigs checked, total coverage {tot_cov / total:.3f}")
    # ---- oracle parity on the native >= 2^32-row index (reference loader: bwt_index.cpp:147-264; loop body: GSAlign.cpp:483-540) ----
    from conftest import assert_stage_equal
    cases = [("contig20_whole_revcomp", qs[20]),                                             # 46 Mb, reverse strand
             ("contig18_whole_forward", qs[18]),                                             # 58 Mb, forward strand
             ("contig0_piece_revcomp", synth.revcomp(np.ascontiguousarray(qs[0][30000000:50000000]))),   # reverse strand of chr1: reference positions ~ 2G - 50 Mb > 2^32
             ("contig7_piece", np.ascontiguousarray(qs[7][5000000:6500000])),
             ("contig22_head", np.ascontiguousarray(qs[22][0:3000001])),
             ("contig20_piece", np.ascontiguousarray(qs[20][300000:1299999])),
             ("contig12_one_chunk", np.ascontiguousarray(qs[12][40000000:40010000]))]
    if scale < 1.0:
        cases = [(n, q[:max(10000, int(q.size * scale))]) for n, q in cases]
    t = time.time()
    if op.have_ref():
        qfa, npz = os.path.join(tmp, "parity_q.fa"), os.path.join(tmp, "parity.npz")
        synth.write_fasta(qfa, cases)
        op.ref_dump_subprocess(os.path.join(tmp, "r"), qfa, npz, PRM, upto=8)      # the real reference, one thread, a process of its own
        want_all = np.load(npz); kind = "real reference (libgsref)"
        want = lambda ci: {k[len(f"c{ci}_"):]: want_all[k] for k in want_all.files if k.startswith(f"c{ci}_")}
    else:
        ora = op.Oracle(idx, PRM); kind = "CPU restatement (oracle/gsa_oracle.cpp)"
        def want(ci):
            ora.set_query(cases[ci][1]); return ora.dump_stages(8)
    print(f"oracle side: {kind}, {sum(q.size for _, q in cases)} bp in {time.time() - t:.0f} s", flush=True)
    hi = rev = 0
    for ci, (name, q) in enumerate(cases):
        w = want(ci)
        g0.set_query(q)
        assert_stage_equal(g0.dump_stages(8), w)
        hi += int((w["s8_f_rpos"] >= 2 ** 32).sum()); rev += int((w["s8_b_bdir"] == 0).sum())
        assert w["s8_b_score"].size > 0 or q.size <= 10000, name
        print(f"  {name}: {q.size} bp, {w['s1_qpos'].size} seeds, {w['s8_b_score'].size} blocks, {w['s8_f_qpos'].size} records, {w['s8_aln1'].size} string bytes: all stage dumps identical", flush=True)
    assert scale < 1.0 or (hi > 0 and rev > 0), (hi, rev)
    print(f"ORACLE PARITY OK on the native wide index: {len(cases)} contigs vs the {kind}; {hi} records at reference positions >= 2^32, {rev} reverse-strand blocks")
    # ---- all 24 contigs against the real reference: stage 1 (seeds, groups) and stage 8 (blocks, records, both gapped-string pools), bit for bit ----
    if ref_jobs:
        n_seeds = n_blocks = n_bytes = 0
        for own, npz, proc in ref_jobs:
            rc = proc.wait(timeout=900)
            assert rc == 0, f"reference process for contigs {own} failed ({rc})"
            with np.load(npz) as z:
                for k, ci in enumerate(own):
                    w = {key[len(f"c{k}_"):]: z[key] for key in z.files if key.startswith(f"c{k}_")}
                    g0.set_query(pinned[ci])
                    assert_stage_equal(g0.dump_stages(8), w, stages=(1, 8))
                    n_seeds += int(w["s1_qpos"].size); n_blocks += int(w["s8_b_score"].size); n_bytes += int(w["s8_aln1"].size)
            os.remove(npz)
        print(f"WHOLE GENOME == real reference: all {len(qs)} contigs ({total} bp) at stages 1 and 8: {n_seeds} seeds, {n_blocks} blocks, {n_bytes} gapped-string bytes per side identical "
              f"({len(ref_jobs)} reference processes, {time.time() - t_ref0:.0f} s after their start)", flush=True)
    # short contigs against the human-sized index, bundled (the PosDiff stride of a bundle is ~ 2G = 6.2 G here: the seed key's width
    # and the PosDiff-sort path at their real size): pieces of five query chromosomes, one reverse-complemented, in one pass == one by one
    pieces = [np.ascontiguousarray(qs[ci][o:o + ln]) for ci, o, ln in ((0, 1000000, 2000000), (7, 5000000, 1500000), (20, 300000, 999999), (22, 0, 3000001), (12, 40000000, 10000))]
    bun = g0.align_bundle(pieces)
    for k, pc in enumerate(pieces):
        alone = g0.align_contig(pc)
        for key in ("blocks", "frags"):
            assert np.array_equal(bun[k][key], alone[key]), (k, key)
        da, db = capi.result_as_dump(bun[k], with_aln=True), capi.result_as_dump(alone, with_aln=True)
        assert np.array_equal(da["aln1"], db["aln1"]) and np.array_equal(da["aln2"], db["aln2"]), (k, "strings")
        assert int(alone["blocks"]["aln_len"].sum()) > 0.5 * pc.size or k == 2, k      # (piece 2 lies on the reverse-complemented contig 20)
    print(f"bundle of {len(pieces)} short contigs against the {idx.G / 1e9:.2f} Gbp index == one by one")
    g1.close(); g0.close()
    print("HUMAN SCALE PROBE OK")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
