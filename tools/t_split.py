"""Host-side timing of one contig with and without the stage timers (hipEvents)."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gsalign_amd import capi
tmp = '/tmp/gsa_split'; os.makedirs(tmp, exist_ok=True)
px, idx, qry = bench.build_workload(tmp, 5_000_000, 0.02, 0, 1)
gpu = capi.Aligner(idx, device=0)
for prof in (False, True, False, True):
    gpu.set_profiling(prof)
    for _ in range(3):
        gpu.set_query(qry); gpu.run_to(8); gpu.block_records()
    ts = []
    for _ in range(30):
        gpu.set_query(qry); torch.cuda.synchronize()
        t0 = time.perf_counter(); gpu.run_to(8); t1 = time.perf_counter(); ts.append(t1 - t0)
    print("profiling %s: run_to(8) median %.3f ms  min %.3f ms" % (prof, 1e3 * np.median(ts), 1e3 * min(ts)))
