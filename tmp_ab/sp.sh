cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/sp -o d -- python tmp_ab/small_probe.py > gpurun_out/sp.log 2>&1
python - <<EOF2
import sqlite3
db=sqlite3.connect("gpurun_out/sp/d_results.db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if "kernel_dispatch" in t][0]; ks=[t for t in tabs if "kernel_symbol" in t][0]
for r in cur.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%k_dp_%' order by d.start"): print(r[0][:20], round(r[1]/1e3,1))
EOF2
rm -rf gpurun_out/sp; tail -3 gpurun_out/sp.log
