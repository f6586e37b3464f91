#!/usr/bin/env python3
"""Where does a contig's time go when several contexts share the GPU?  From a rocpd kernel trace of bench.py: the operations of every MAIN stream (the
stream that carries k_seed_select) are cut into contigs, and for the last K contigs the wall time between phase boundaries is averaged:
seed (first seed kernel .. k_seed_select end) | chain (.. OpEarlyGaps end) | refine (.. k_leaf_emit end) | extend passes (.. OpClassify end) |
tail (.. k_materialize_large end: small DP kernels, strings, waiting for the striped DP) -- and how long the striped kernels of that contig ran beside it.
    python tools/contig_phases.py results.db [K=96]"""
import collections, sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
scol = "stream_id" if "stream_id" in cols else "queue_id"
ops = [(r[0], r[1], r[2], r[3]) for r in cur.execute(f"select d.start, d.end, s.kernel_name, d.{scol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start")]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 96
main = {o[3] for o in ops if "k_seed_select" in o[2]}
per = collections.defaultdict(list)
for o in ops:
    if o[3] in main: per[o[3]].append(o)
contigs = []
for st, lst in per.items():
    cur_c = None
    for s, e, n, _ in lst:
        if ("k_seed_wg" in n or "k_dense_sweep" in n or ("k_dense_search" in n)) and (cur_c is None or "sel" in cur_c):
            if cur_c and "end" in cur_c: contigs.append(cur_c)
            cur_c = {"t0": s}
        if cur_c is None: continue
        if "k_seed_select" in n: cur_c["sel"] = e
        elif "OpEarlyGaps" in n: cur_c["chain"] = e
        elif "k_leaf_emit" in n: cur_c["refine"] = e
        elif "OpClassify" in n: cur_c["passes"] = e
        elif "k_materialize_large" in n: cur_c["end"] = e
    if cur_c and "end" in cur_c: contigs.append(cur_c)
contigs = [c for c in contigs if all(k in c for k in ("sel", "chain", "refine", "passes", "end"))]
contigs.sort(key=lambda c: c["t0"]); contigs = contigs[-K - 4:-4]
def avg(f): return sum(f(c) for c in contigs) / max(1, len(contigs)) / 1e3
print(f"{len(contigs)} contigs on {len(main)} main streams; mean wall time per contig {avg(lambda c: c['end'] - c['t0']):.0f} us:")
for name, f in (("seed (.. k_seed_select)", lambda c: c["sel"] - c["t0"]), ("chain (.. OpEarlyGaps)", lambda c: c["chain"] - c["sel"]), ("refine (.. k_leaf_emit)", lambda c: c["refine"] - c["chain"]),
                ("extend passes (.. OpClassify)", lambda c: c["passes"] - c["refine"]), ("tail (.. k_materialize_large)", lambda c: c["end"] - c["passes"])):
    print(f"  {name:34s} {avg(f):9.0f} us")
# kernel time on the main stream inside each phase vs the phase's wall time: the rest is waiting (launch gaps, events of other streams, host look-ins)
busy = collections.defaultdict(float)
for st, lst in per.items():
    for s, e, n, _ in lst:
        for c in contigs:
            if c["t0"] <= s < c["end"]:
                ph = "seed" if s < c["sel"] else "chain" if s < c["chain"] else "refine" if s < c["refine"] else "passes" if s < c["passes"] else "tail"
                busy[ph] += (e - s); break
print("  main-stream kernel time inside the phases (us per contig):", {k: round(v / 1e3 / max(1, len(contigs))) for k, v in busy.items()})
