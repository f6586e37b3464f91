#!/usr/bin/env python3
"""A reference with MORE than 2^31 bases (text = forward + reverse complement > 2^32 rows): the 64-bit instance of the index
builder's suffix sorter and the wide device layout (64-bit dense SA, 32-byte k-mer entries) on their real input, not forced.
GPU box only; needs ~70 GB of host memory and ~60 GB of HBM.   usage: big_index_probe.py [total_Gbp=2.2] [n_chr=10]"""
import os, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gsalign_amd import synth, hostlib, indexio, capi
total = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 2_200_000_000
nchr = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mem_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2**30
print(f"host memory {mem_gb:.0f} GiB", flush=True)
if mem_gb < 110:
    print("not enough host memory for this probe"); sys.exit(0)
tmp = tempfile.mkdtemp(prefix="bigidx_", dir="/tmp")
try:
    t = time.time()
    refs = [(f"chr{i + 1}", synth.fast_genome(total // nchr, 31000 + i)) for i in range(nchr)]
    synth.write_fasta(os.path.join(tmp, "r.fa"), refs)
    print(f"reference: {nchr} x {total // nchr} bp written in {time.time() - t:.0f} s", flush=True)
    t = time.time(); hostlib.build_index(os.path.join(tmp, "r.fa"), os.path.join(tmp, "r")); tb = time.time() - t
    print(f"index built in {tb:.0f} s (64-bit suffix sorter: 2G + 1 = {2 * (total // nchr) * nchr + 1} suffixes)", flush=True)
    t = time.time(); idx = indexio.load_index(os.path.join(tmp, "r")); print(f"index loaded in {time.time() - t:.0f} s, seq_len {idx.seq_len} (>= 2^32: {idx.seq_len >= 2**32})", flush=True)
    t = time.time(); g = capi.Aligner(idx); print(f"gsa_create (upload, dense SA, k-mer table) {time.time() - t:.0f} s", flush=True)
    from test_gpu_parity import _check_result_invariants
    qs = [("fwd", synth.fast_mutate(refs[2][1][5_000_000:55_000_000], 0.01, 77)),
          ("rev_last_chr", synth.revcomp(synth.fast_mutate(refs[nchr - 1][1][-30_000_000:], 0.01, 78)))]
    for name, q in qs:
        t = time.time(); g.align_contig(q); dt = time.time() - t
        r = g.blocks()
        _check_result_invariants(idx, q, r)
        cov = int(r["blocks"]["aln_len"].sum())
        print(f"{name}: {q.size} bp aligned in {dt * 1e3:.1f} ms, {r['blocks'].size} blocks, coverage {cov / q.size:.3f}, chr {sorted(set(r['blocks']['chr'].tolist()))}, invariants ok", flush=True)
        assert cov > 0.9 * q.size
    g.close()
    print("BIG INDEX PROBE OK")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
