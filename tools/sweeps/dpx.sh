# timing experiments on the striped DP (GPU box): rebuild with -DDP_EXP=x and run the shape batches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for x in "$@"; do
rm -rf gsalign_amd/csrc/build; make -C gsalign_amd/csrc -j32 lib EXTRA="$x" > /tmp/mk.log 2>&1 || tail -5 /tmp/mk.log
rocprofv3 --kernel-trace -d gpurun_out/dpb -o d -- python tools/dp_batch_probe.py > gpurun_out/dpb.log 2>&1
python - "$x" <<EOF2
import sqlite3, sys
db=sqlite3.connect("gpurun_out/dpb/d_results.db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if "kernel_dispatch" in t][0]; ks=[t for t in tabs if "kernel_symbol" in t][0]
out=[]
for r in cur.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%k_dp_stripe%' order by d.start"): out.append(round(r[1]/1e3))
print(sys.argv[1], out[1::2])
EOF2
rm -rf gpurun_out/dpb
done
