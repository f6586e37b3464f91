import os, sys, tempfile
sys.path.insert(0, "/root/repo"); os.environ["GSA_DEBUG"]="1"
import bench
from gsalign_amd import capi
wl = bench.WORKLOADS["human"]; tmp = tempfile.mkdtemp()
px, idx, refs = bench.build_reference(tmp, "human", wl, 0, 1)
q = bench.make_queries(wl, refs, 0)[0][0]
g = capi.Aligner(idx); g.align_contig_raw(q); g.align_contig_raw(q)
