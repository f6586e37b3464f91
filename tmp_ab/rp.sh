cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/sp -o d -- python tmp_ab/t1.py > gpurun_out/rp.log 2>&1
python - <<EOF2
import sqlite3
db=sqlite3.connect("gpurun_out/sp/d_results.db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if "kernel_dispatch" in t][0]; ks=[t for t in tabs if "kernel_symbol" in t][0]
rows=list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
last=[i for i,r in enumerate(rows) if 'k_seed_wg' in r[0]][-1]
t0=rows[last][1]
for r in rows[last:]:
    if (r[2]-r[1])>150e3 or 'k_dp_' in r[0]: print(f"{(r[1]-t0)/1e3:9.1f} +{(r[2]-r[1])/1e3:8.1f} {r[0][:40]}")
EOF2
rm -rf gpurun_out/sp; grep "align_contig_raw" gpurun_out/rp.log | tail -4
