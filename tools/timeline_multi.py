#!/usr/bin/env python3
"""Kernel-trace timeline of SEVERAL contexts in flight (rocpd sqlite from rocprofv3 --kernel-trace):
which queue / stream every operation ran on, how much of a steady-state window the GPU ran at least one
kernel, per-stream busy time and the distribution of the gaps in front of the operations of each stream.

    rocprofv3 --kernel-trace -d gpurun_out/tlm -o t -- python bench.py --workload ecoli --inflight 3 --steps 24 --warmup 6 --extra "" --no-cpu-baseline
    python tools/timeline_multi.py gpurun_out/tlm/t_results.db [window_us] [dump]
"""
import collections
import sqlite3
import sys


def main(path, window_us=4000.0, dump=False, at_contig=18):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    print("# columns:", ", ".join(cols))
    qcol = "queue_id" if "queue_id" in cols else "0"; scol = "stream_id" if "stream_id" in cols else "0"; tcol = "tid" if "tid" in cols else "0"
    ops = [(r[0], r[1], r[2].split("(")[0][:44].replace(".kd", ""), r[3], r[4], r[5], r[6]) for r in cur.execute(
        f"select d.start, d.end, s.kernel_name, d.{qcol}, d.{scol}, d.{tcol}, d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start")]
    if not ops: print("no dispatches"); return
    seeds = [o for o in ops if "k_seed_wg" in o[2] or "k_dense_search" in o[2] and "resolve" not in o[2]]
    t_first, t_last = seeds[0][0], seeds[-1][0]
    n_contig = len([o for o in ops if "k_seed_select" in o[2]])
    print(f"# {len(ops)} dispatches, {n_contig} contigs (k_seed_select launches), seed launches span {(t_last - t_first) / 1e6:.2f} ms -> {(t_last - t_first) / 1e3 / max(1, n_contig - 1):.1f} us per contig")
    sel = [o for o in ops if "k_seed_select" in o[2]]
    mid = sel[min(len(sel) - 1, at_contig)][0]              # (the bench's first timed region: warm-up + steps / 2)
    w0, w1 = mid - int(window_us * 500), mid + int(window_us * 500)
    win = [o for o in ops if o[1] > w0 and o[0] < w1]
    # union busy
    busy = 0; cur_end = w0; conc = []
    ev = []
    for s, e, *_ in win:
        s = max(s, w0); e = min(e, w1)
        ev.append((s, 1)); ev.append((e, -1))
        if e > cur_end: busy += e - max(s, cur_end); cur_end = e
    ev.sort(); depth = 0; prev = w0; hist = collections.Counter()
    for t, d in ev:
        hist[depth] += t - prev; prev = t; depth += d
    hist[depth] += w1 - prev
    print(f"# window {window_us:.0f} us around contig {at_contig}: {len(win)} ops, GPU runs >= 1 kernel {100.0 * busy / (w1 - w0):.1f} % of it; kernels in flight: " +
          ", ".join(f"{k}: {100.0 * v / (w1 - w0):.0f}%" for k, v in sorted(hist.items())))
    # per queue / stream
    for key_i, label in ((3, "queue"), (4, "stream")):
        per = collections.defaultdict(list)
        for o in win: per[o[key_i]].append(o)
        print(f"# per {label}:")
        for k, lst in sorted(per.items(), key=lambda kv: str(kv[0])):
            b = sum(min(e, w1) - max(s, w0) for s, e, *_ in lst)
            gaps = [lst[i + 1][0] - max(x[1] for x in lst[:i + 1][-4:]) for i in range(len(lst) - 1)]
            gaps = sorted(g / 1e3 for g in gaps if g > 0)
            med = gaps[len(gaps) // 2] if gaps else 0.0
            print(f"#   {label} {k}: {len(lst)} ops, busy {100.0 * b / (w1 - w0):.0f} %, gaps: n {len(gaps)} median {med:.1f} us p90 {gaps[int(len(gaps) * 0.9)] if gaps else 0:.1f} us sum {sum(gaps):.0f} us")
    # kernels by time inside the window
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n, *_ in win: agg[n][0] += 1; agg[n][1] += e - s
    print("# top kernels in the window (occupancy-time):")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"#   {n:44s} {c:4d} x {t / 1e3 / c:8.1f} us = {t / 1e3:8.1f} us")
    if dump:
        for s, e, n, q, st, tid, gx in win:
            print(f"{(s - w0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  q{q} s{st} t{tid % 1000} g{gx:<8d} {n}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 4000.0, len(sys.argv) > 3)
