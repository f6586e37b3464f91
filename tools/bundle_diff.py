#!/usr/bin/env python3
"""Flake hunt for bundles (GPU box): every result of gsa_align_many against the result the contig gets ALONE; for a result that
differs, which gap records hold wrong strings (by string length = kind of gap, by position in the result).
usage: bundle_diff.py [workload=ecoli] [rounds=100]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ctypes as C
import bench
from gsalign_amd import capi
name = sys.argv[1] if len(sys.argv) > 1 else "ecoli"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
wl = dict(bench.WORKLOADS[name])
tmp = os.environ.get("BDIFF_TMP") or tempfile.mkdtemp(prefix="bdiff_")
px, idx, refs = bench.build_reference(tmp, name, wl, 0, 1)
contigs = [c for gq in bench.make_queries(wl, refs, 0) for c in gq]
g0 = capi.Aligner(idx, **wl["params"]); ctxs = [g0, g0.clone(), g0.clone()]
pinned = [g0.pinned_copy(c) for c in contigs]
def grab(res):
    nb, nf, na = res.n_blocks, res.n_frags, res.n_aln
    return (np.frombuffer(C.string_at(res.blocks, nb * 40), np.uint8).copy(), np.frombuffer(C.string_at(res.recs, nf * 16), np.int32).reshape(-1, 4).copy(),
            np.frombuffer(C.string_at(res.aln1, na), np.uint8).copy(), np.frombuffer(C.string_at(res.aln2, na), np.uint8).copy())
alone = {}
def on_alone(ci, res): alone[ci] = grab(res); return 0
CHILD = os.environ.get("BDIFF_CHILD") == "1"
if not CHILD:
    os.environ["GSA_BUNDLE_CONTIG"] = "0"      # (read once per process by the library) -- the results the contigs get alone
    for k, p in enumerate(pinned):
        capi.align_many([g0], [p], lambda ci, res, k=k: on_alone(k, res))
else:
    import pickle
    alone = pickle.load(open(os.path.join(tmp, "alone.pkl"), "rb"))
bad = []
def on_result(ci, res):
    k = ci % len(contigs); cur = grab(res); ref = alone[k]
    if not (np.array_equal(ref[0], cur[0]) and np.array_equal(ref[1], cur[1])): bad.append((ci, "blocks/recs differ")); return 0
    if np.array_equal(ref[2], cur[2]) and np.array_equal(ref[3], cur[3]): return 0
    recs = cur[1]; gaps = np.nonzero(recs[:, 0] < 0)[0]
    off = recs[gaps, 3].astype(np.int64) & 0xffffffff; ln = recs[gaps, 2].astype(np.int64)
    d = (ref[2] != cur[2]) | (ref[3] != cur[3])
    cs = np.concatenate(([0], np.cumsum(d)))
    nbad = cs[off + ln] - cs[off]
    wrong = np.nonzero(nbad > 0)[0]
    cover = np.zeros(len(cur[2]) + 1, np.int32); np.add.at(cover, off, 1); np.add.at(cover, off + ln, -1); covered = np.cumsum(cover[:-1]) > 0
    bad.append((ci, f"{int(d.sum())} bytes differ, {int((d & covered).sum())} of them inside a gap record's string ({int(covered.sum())} of {len(covered)} pool bytes belong to a record); {len(wrong)} of {len(gaps)} gap strings wrong"))
    return 0
# (the environment variable is read once per process: the bundled run happens in a child process)
if not CHILD:
    import pickle, subprocess
    pickle.dump(alone, open(os.path.join(tmp, "alone.pkl"), "wb"))
    env = dict(os.environ, BDIFF_CHILD="1", BDIFF_TMP=tmp); env.pop("GSA_BUNDLE_CONTIG", None)
    sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__), name, str(rounds)], env=env).returncode)
capi.align_many(ctxs, pinned * rounds, on_result)
print(f"{name}: {len(contigs) * rounds} bundled results against the results alone: {len(bad)} differ")
for b in sorted(bad)[:12]: print("  result", b[0], "(contig", b[0] % len(contigs), "):", b[1])
print("  differing result numbers:", sorted(x[0] for x in bad)[:80])
