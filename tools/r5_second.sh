#!/bin/bash
# round 5, second GPU call: where the seeds differ on the native wide index; the suite with k_walk_chain; early / late striped-job counts
cd "$GRAFT_REPO_ROOT"
( time timeout 1200 python tests/human_scale_diag.py ) > gpurun_out/r5_diag.log 2>&1
tail -60 gpurun_out/r5_diag.log
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "not config5" ) > gpurun_out/r5_gputest2.log 2>&1
tail -8 gpurun_out/r5_gputest2.log
( GSA_WALK_CHAIN_MIN=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or drop_in or complex or degenerate" ) > gpurun_out/r5_gputest2b.log 2>&1
tail -4 gpurun_out/r5_gputest2b.log
python - > gpurun_out/r5_early_late.txt 2>&1 <<'P'
import sys, time; sys.path.insert(0, ".")
import numpy as np
from gsalign_amd import synth, hostlib, indexio, capi
import tempfile, os
tmp = tempfile.mkdtemp()
r = synth.fast_genome(250_000_000, 11000); synth.inject_repeats(r, 11000)
synth.write_fasta(tmp + "/r.fa", [("chr1", r)]); hostlib.build_index(tmp + "/r.fa", tmp + "/r")
idx = indexio.load_index(tmp + "/r"); g = capi.Aligner(idx)
q = synth.fast_mutate(r, 0.01, 7000)
for rep in range(3):
    t = time.time(); g.align_contig(q); dt = time.time() - t
    st = g.seed_stats(); c = g.counters()
    print(f"250 Mb contig: {dt*1e3:.1f} ms; large gaps launched early {int(st[6])}, large jobs in the late launch {int(st[7])}, DP jobs {int(c[5])}")
g.set_profiling(True); g.align_contig(q); print("stage ms", g.timings())
P
cat gpurun_out/r5_early_late.txt
