#!/usr/bin/env python3
"""VERDICT r2 item 7 at size: a 250 Mb reference with the adversarial injection (csrc/host/synth.cpp: repeat families with a copy-number
spectrum up to 10^5, microsatellites, two Mb-long N runs, soft-masked blocks) vs a 1 %-diverged query -- how many chunks the speculative
seed kernel hands to the dense search, stage times of one context alone, and the throughput of two contexts.  GPU box.
    python tools/adversarial_probe.py [genome_len] > profiles/archive/r03_adversarial_probe.txt"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gsalign_amd import synth, hostlib, indexio, capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
tmp = os.environ.get("GSA_PROBE_KEEP") or tempfile.mkdtemp(prefix="advprobe_")
os.makedirs(tmp, exist_ok=True)
kinds = [k for k in ("repeat-stress (bench workload)", "adversarial") if len(sys.argv) < 3 or k[:6] in sys.argv[2]]
for kind in kinds:
    r = synth.fast_genome(n, 11000)
    copies = synth.inject_repeats(r, 11000) if kind.startswith("repeat") else synth.inject_adversarial(r, 11000)
    px = os.path.join(tmp, f"{kind[:6]}_{n}")
    t = time.time()
    if not os.path.exists(px + ".done"):
        synth.write_fasta(px + ".fa", [("chr1", r)]); hostlib.build_index(px + ".fa", px); open(px + ".done", "w").close()
    tb = time.time() - t
    idx = indexio.load_index(px)
    qs = [synth.fast_mutate(r, 0.01, 7000 + k) for k in range(2)]
    g = capi.Aligner(idx); g2 = g.clone()
    g.set_profiling(True)
    for rep in range(2):
        g.align_contig(qs[0])
    tm = g.timings(); st = g.seed_stats(); c = g.counters(); res = g.raw_result()
    print(f"== {kind}: {n} bp, {copies} repeat copies, index build {tb:.0f} s")
    print(f"   chunks {(qs[0].size + 9999) // 10000}, redone by the dense search {int(st[1])}; hits {int(c[2])}, DP jobs {int(c[5])}, DP cells {int(c[4])}; blocks {res.n_blocks}, records {res.n_frags}")
    print("   one context alone, ms: seed search %.2f | locate %.2f | sort %.2f | chain %.2f | refine %.2f | extend %.2f" % tuple(float(x) for x in tm[:6]))
    g.set_profiling(False)
    dev = [g.device_copy(q) for q in qs]
    capi.align_many([g, g2], dev * 2, in_order=True)
    t = time.time(); capi.align_many([g, g2], dev * 6, in_order=True); dt = time.time() - t
    print(f"   two contexts, contigs resident: {12 * qs[0].size / dt / 1e9:.2f} Gbp/s ({1000 * dt / 12:.2f} ms per contig)", flush=True)
    g2.close(); g.close()
