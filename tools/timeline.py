#!/usr/bin/env python3
"""Per-step GPU timeline from a rocprofv3 rocpd database (kernel + memory-copy trace):
busy time, idle gaps and the ops of the last bench step.

    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl -o tl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/timeline.py gpurun_out/tl/tl_results.db
"""
import sqlite3
import sys


def main(path, verbose=False):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
    mc = [t for t in tabs if "memory_copy" in t]
    ops = [(r[0], r[1], r[2].split("(")[0][:48]) for r in cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id")]
    if mc:
        cols = [r[1] for r in cur.execute(f"pragma table_info({mc[0]})")]
        sz = "size" if "size" in cols else None
        for r in cur.execute(f"select start, end{', ' + sz if sz else ''} from {mc[0]}"):
            ops.append((r[0], r[1], f"<copy {r[2] if sz else ''}>"))
    ops.sort()
    # last step = ops after the last k_seed_wg launch
    idx = [i for i, o in enumerate(ops) if "k_seed_wgILb0" in o[2] or "k_dense_searchILb" in o[2] or ("k_dense_sweep" in o[2])]      # (not the accounting build k_seed_wg<true, .>)
    if idx:      # a pass may launch the speculative kernel AND a dense one: the step starts at the first seed launch behind the last k_seed_select in front of it
        sel = [i for i, o in enumerate(ops) if "k_seed_select" in o[2] and i < idx[-1]]
        first = [i for i in idx if not sel or i > sel[-1]]
        idx = [first[0]] if first else idx
    if not idx:
        print("no seed kernel found"); return
    step = ops[idx[-1]:]
    t0, t1 = step[0][0], max(o[1] for o in step)
    busy = 0; cur_end = t0; gaps = []
    for s, e, n in step:
        if s > cur_end: gaps.append((s - cur_end, n))
        if e > cur_end: busy += e - max(s, cur_end); cur_end = e
    print(f"last step: {len(step)} GPU ops, span {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us in {len(gaps)} gaps")
    gaps.sort(reverse=True)
    print("largest gaps (us, before op):", ", ".join(f"{g / 1e3:.1f}:{n}" for g, n in gaps[:12]))
    if verbose:
        prev = t0
        for s, e, n in step:
            print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  gap {max(0, s - prev) / 1e3:6.1f}  {n}")
            prev = max(prev, e)


if __name__ == "__main__":
    main(sys.argv[1], len(sys.argv) > 2)
