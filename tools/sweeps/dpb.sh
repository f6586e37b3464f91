cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ksw2 or 1500000-1-0.001 or fallback" 2>&1 | tail -2
rocprofv3 --kernel-trace -d gpurun_out/dpb -o d -- python tools/dp_batch_probe.py > gpurun_out/dpb.log 2>&1
python - <<EOF2
import sqlite3
db=sqlite3.connect("gpurun_out/dpb/d_results.db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if "kernel_dispatch" in t][0]; ks=[t for t in tabs if "kernel_symbol" in t][0]
out=[]
for r in cur.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%k_dp_%' order by d.start"): out.append((r[0][4:16], round(r[1]/1e3,1)))
print(out)
EOF2
rm -rf gpurun_out/dpb
