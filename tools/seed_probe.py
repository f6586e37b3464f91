#!/usr/bin/env python3
"""Seed-search kernel under repeats: time + in-kernel counters per reference variant (GPU box).
usage: seed_probe.py [genome_len] [variants: none,family,tandem,both,sen]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gsalign_amd import synth, hostlib, indexio, capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
variants = (sys.argv[2] if len(sys.argv) > 2 else "none,family,tandem,both").split(",")
tmp = os.environ.get("GSA_PROBE_KEEP") or tempfile.mkdtemp(prefix="seedprobe_")      # (GSA_PROBE_KEEP: reuse the index between calls)
os.makedirs(tmp, exist_ok=True)
for v in variants:
    r = synth.fast_genome(n, 11000)
    div, prm = 0.01, {}
    if v == "family": synth.inject_repeats(r, 11000, tandem_unit=0, tandem_copies=0)
    elif v == "tandem": synth.inject_repeats(r, 11000, frac=0.0)
    elif v == "both": synth.inject_repeats(r, 11000)
    elif v == "sen": div, prm = 0.02, dict(sen=1, clr=50)
    px = os.path.join(tmp, v)
    px = px + f"_{n}"
    t = time.time()
    if not os.path.exists(px + ".done"):
        synth.write_fasta(px + ".fa", [("chr1", r)]); hostlib.build_index(px + ".fa", px); open(px + ".done", "w").close()
    tb = time.time() - t
    idx = indexio.load_index(px)
    q = synth.fast_mutate(r, div, 7000)
    g = capi.Aligner(idx, **prm)
    g.set_profiling(True)
    for rep in range(2):
        g.set_query(q); g.run_to(1)
    tm = g.timings(); c = g.counters(); st = g.seed_stats()
    if os.environ.get("SEED_STATS"):      # (library built with -DSEED_STATS: sums over the waves of the LAST launch instead of maxima / timers)
        it, fm_any, fm_only, act, fm = (int(st[0]), int(st[2]), int(st[3]), int(st[4]), int(st[5]))
        print(f"   SEED_STATS: wave-iterations {it}, with an FM lane {fm_any} ({fm_any / max(1, it):.2f}), FM lanes only {fm_only} ({fm_only / max(1, it):.2f}), lanes active per iteration {act / max(1, it):.1f}, FM lanes per iteration {fm / max(1, it):.2f}")
    print(f"   seed stats: resolver rounds max {int(st[0])}, dense chunks {int(st[1])}, wave iterations max {int(st[2])}; slowest chunk: round 1 {st[3] * 0.01:.1f} us, resolver {st[4] * 0.01:.1f} us, total {st[5] * 0.01:.1f} us")
    print(f"== {v}: n={n} index build {tb:.1f}s  seed_search {tm[0]:.3f} ms locate {tm[1]:.3f} sort {tm[2]:.3f}  hits {int(c[2])} occ_read {int(c[7])}", flush=True)
    g.set_query(q); g.run_to(8); tm = g.timings()
    print("   stages:", [round(float(x), 3) for x in tm], flush=True)
    g.close()
