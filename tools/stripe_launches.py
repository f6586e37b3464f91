#!/usr/bin/env python3
"""The striped-DP launches of a kernel trace (rocpd sqlite): grid, LDS, duration -- which size classes ran."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(cur.execute(f"select d.start, d.end, s.kernel_name, d.grid_size_x, d.workgroup_size_x, d.group_segment_size from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%k_dp_%' order by d.start"))
t0 = rows[0][0] if rows else 0
for s, e, n, g, w, l in rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -24:]:
    print(f"{(s - t0) / 1e3:12.1f} +{(e - s) / 1e3:9.1f} us  workgroups {g // max(1, w):7d} x {w:4d}  lds {l:6d}  {n.split('(')[0][:40]}")
