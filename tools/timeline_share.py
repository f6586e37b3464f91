#!/usr/bin/env python3
"""Whose time is a multi-context step?  From a rocpd kernel trace (rocprofv3 --kernel-trace of bench.py with several contexts in flight): over a
steady-state window, how long k kernels were in flight at once, and every kernel's SHARE of the window -- each instant is split evenly among the
kernels running at it, so the shares add up to the busy time: a kernel that runs alone for 1 ms owns 1 ms, four that overlap for 1 ms own 0.25 ms each.
    python tools/timeline_share.py results.db [contigs_in_window=100]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
ops = [(r[0], r[1], r[2].split("(")[0].replace(".kd", "")) for r in cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start")]
# the window: the last K contigs of the trace (K k_seed_select launches from the end, the very last few left out): the timed steps of bench.py
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sel = [o for o in ops if "k_seed_select" in o[2]]
w0 = sel[max(0, len(sel) - K)][0]; w1 = sel[max(0, len(sel) - 1 - max(4, K // 20))][0]
ev = []
for i, (s, e, n) in enumerate(ops):
    s, e = max(s, w0), min(e, w1)
    if e > s: ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
live = set(); last = w0; share = collections.defaultdict(float); depth = collections.defaultdict(float); alone = collections.defaultdict(float)
def short(n):
    import re
    m = re.search(r"k_lb_passILi\d+E\d+(Op[A-Za-z]+)", n)
    if m: return "lb:" + m.group(1)
    m = re.search(r"_Z\d+(k_[a-z_0-9]+)", n)
    return m.group(1) if m else n[:40]
for t, d, i in ev:
    if t > last and live:
        dt = t - last; depth[len(live)] += dt
        for j in live: share[short(ops[j][2])] += dt / len(live)
        if len(live) == 1: alone[short(ops[next(iter(live))][2])] += dt
    elif t > last: depth[0] += t - last
    last = t
    (live.add if d > 0 else live.discard)(i)
W = (w1 - w0) / 1e3
print(f"window {W:.0f} us; kernels in flight: " + ", ".join(f"{k}: {100 * v / 1e3 / W:.0f}%" for k, v in sorted(depth.items())))
print("share of the window (each instant split among the kernels running at it) | of which running ALONE")
for n, v in sorted(share.items(), key=lambda x: -x[1])[:28]:
    print(f"  {n:34s} {100 * v / 1e3 / W:6.2f} %   alone {100 * alone.get(n, 0) / 1e3 / W:5.2f} %")
