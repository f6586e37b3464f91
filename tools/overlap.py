#!/usr/bin/env python3
"""How well do the in-flight contexts overlap?  From a rocprofv3 rocpd database (kernel + memory-copy trace): for the
last `frac` of the trace, the span, the union of GPU-busy time, the summed kernel time, and the biggest ops.
    python tools/overlap.py results.db [frac=0.3]"""
import sqlite3, sys
from collections import defaultdict
path = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
db = sqlite3.connect(path); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
mc = [t for t in tabs if "memory_copy" in t]
ops = [(r[0], r[1], r[2].split("(")[0][:40], r[3]) for r in cur.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id")]
cp = []
if mc:
    cols = [r[1] for r in cur.execute(f"pragma table_info({mc[0]})")]
    print("memcopy cols", cols)
    sz = "size" if "size" in cols else None
    for r in cur.execute(f"select start, end{', ' + sz if sz else ''} from {mc[0]}"):
        cp.append((r[0], r[1], r[2] if sz else 0))
t1 = max(o[1] for o in ops); t0a = min(o[0] for o in ops); t0 = t1 - (t1 - t0a) * frac
sel = sorted(o for o in ops if o[0] >= t0)
busy = 0; ce = t0
for s, e, n, q in sel:
    if e > ce: busy += e - max(s, ce); ce = e
tot = sum(e - s for s, e, n, q in sel)
print(f"window {(t1 - t0) / 1e6:.1f} ms: GPU busy (union of kernels) {busy / 1e6:.1f} ms, summed kernel time {tot / 1e6:.1f} ms, queues {sorted(set(q for *_, q in sel))}")
per = defaultdict(float)
for s, e, n, q in sel: per[n] += e - s
for n, v in sorted(per.items(), key=lambda kv: -kv[1])[:10]: print(f"   {n:42s} {v / 1e6:8.2f} ms")
csel = [c for c in cp if c[0] >= t0]
if csel:
    print(f"copies in window: {len(csel)}, summed {sum(e - s for s, e, _ in csel) / 1e6:.1f} ms, bytes {sum(b for *_, b in csel) / 1e6:.0f} MB")
    for s, e, b in sorted(csel, key=lambda c: -(c[1] - c[0]))[:6]: print(f"   copy {b / 1e6:8.1f} MB {(e - s) / 1e3:9.1f} us -> {b / max(1, e - s):.1f} GB/s")
# seed kernel launches in the window: how far apart do they start, do they overlap with stripes of another context?
seeds = [(s, e) for s, e, n, q in sel if "k_seed_wg" in n]
print("seed kernel starts (ms from window start):", [round((s - t0) / 1e6, 1) for s, e in seeds][:20])
print("seed kernel durations (ms):", [round((e - s) / 1e6, 1) for s, e in seeds][:20])
