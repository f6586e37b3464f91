import sys, time, os, numpy as np
sys.path.insert(0, '/root/repo')
import torch
import bench
from gsalign_amd import capi
tmp='/tmp/gsa_split'; os.makedirs(tmp, exist_ok=True)
px, idx, qry = bench.build_workload(tmp, 5_000_000, 0.02, 0, 1)
gpu = capi.Aligner(idx, device=0)
for _ in range(3):
    gpu.set_query(qry); gpu.run_to(8); gpu.block_records()
ts=[[],[],[],[]]
for _ in range(20):
    gpu.set_query(qry); torch.cuda.synchronize()
    t0=time.perf_counter(); gpu.run_to(3); t1=time.perf_counter(); gpu.run_to(6); t2=time.perf_counter(); gpu.run_to(8); t3=time.perf_counter(); r=gpu.block_records(); t4=time.perf_counter()
    ts[0].append(t1-t0); ts[1].append(t2-t1); ts[2].append(t3-t2); ts[3].append(t4-t3)
print("run_to(3) %.3f  4-6 %.3f  7-8 %.3f  block_records %.3f ms" % tuple(1e3*np.median(x) for x in ts))
ts=[]
for _ in range(20):
    gpu.set_query(qry); torch.cuda.synchronize()
    t0=time.perf_counter(); gpu.run_to(8); t1=time.perf_counter(); ts.append(t1-t0)
print("run_to(8) alone %.3f ms" % (1e3*np.median(ts)))
