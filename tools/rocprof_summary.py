#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel summary text we
commit under profiles/ (the .db itself is scratch under gpurun_out/).

    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/archive/r01_xxx.txt
"""
import sqlite3
import sys


def main(path, top=60):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size), min(d.group_segment_size) "
         f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"# {len(rows)} kernels, {sum(r[1] for r in rows)} dispatches, total kernel time {total / 1e6:.3f} ms")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s} {'lds_min':>7s}")
    print("# (lds = the largest LDS allocation among the kernel's launches, lds_min the smallest: the striped DP runs in size classes)")
    for r in rows[:top]:
        name = r[0].replace(".kd", "")
        print(f"{name[:70]:70s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} {100.0 * r[2] / total:6.2f} {r[6] or 0:5d} {r[7] or 0:5d} {r[8] or 0:7d} {r[9] or 0:7d} {r[10] or 0:7d}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
