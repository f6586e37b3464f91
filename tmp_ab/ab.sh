cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
[ -n "$FUZZ" ] && python tools/dp_fuzz.py 4000 11 2>&1 | tail -4
for v in "$@"; do
cp tmp_ab/$v.so gsalign_amd/lib/libgsa_hip.so
rocprofv3 --kernel-trace -d gpurun_out/dpb_$v -o d -- python tools/dp_batch_probe.py > gpurun_out/dpb_$v.log 2>&1
python - <<EOF2
import sqlite3
db=sqlite3.connect("gpurun_out/dpb_$v/d_results.db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if "kernel_dispatch" in t][0]; ks=[t for t in tabs if "kernel_symbol" in t][0]
out=[]
for r in cur.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%k_dp_stripe%' order by d.start"): out.append(round(r[1]/1e3,1))
print("$v", out[1::2])
EOF2
rm -rf gpurun_out/dpb_$v
done
