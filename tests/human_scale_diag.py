#!/usr/bin/env python3
"""Diagnostic (GPU box, tests/ because it loads oracle/): on the native >= 2^32-row index, where do the GPU's stage-1 seeds differ from the real
reference's?  Pieces from several places, three seed modes, the differing seeds listed with what the leaf operator says about their starts.
usage: human_scale_diag.py [scale=1.0]"""
import os, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gsalign_amd import synth, hostlib, indexio, capi
from oracle import oracle_py as op
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
MB = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57]
lens = [int(m * 1e6 * scale) for m in MB]
tmp = tempfile.mkdtemp(prefix="humand_", dir="/tmp")
try:
    t = time.time(); refs = []
    for i, n in enumerate(lens):
        r = synth.fast_genome(n, 41000 + i); synth.inject_repeats(r, 41000 + i); refs.append((f"chr{i + 1}", r))
    synth.write_fasta(os.path.join(tmp, "r.fa"), refs)
    hostlib.build_index(os.path.join(tmp, "r.fa"), os.path.join(tmp, "r")); print(f"index built in {time.time() - t:.0f} s", flush=True)
    idx = indexio.load_index(os.path.join(tmp, "r")); G = idx.G
    def qof(ci): return synth.fast_mutate(refs[ci][1], 0.01, 51000 + ci)
    q20 = synth.revcomp(qof(20)); q0 = qof(0); q7 = qof(7); q22 = qof(22)
    cases = [("c20_rev_0_3M", q20[:3000000]), ("c20_rev_20M_23M", q20[20000000:23000000]), ("c0_fwd_1M_4M", q0[1000000:4000000]), ("c0_rev_30M_33M", synth.revcomp(np.ascontiguousarray(q0[30000000:33000000]))),
             ("c7_fwd", q7[5000000:8000000]), ("c22_fwd_head", q22[:3000000]), ("c22_rev_tail", synth.revcomp(np.ascontiguousarray(q22[-3000000:])))]
    cases = [(n, np.ascontiguousarray(q)) for n, q in cases]
    qfa, npz = os.path.join(tmp, "d_q.fa"), os.path.join(tmp, "d.npz")
    synth.write_fasta(qfa, cases)
    t = time.time(); op.ref_dump_subprocess(os.path.join(tmp, "r"), qfa, npz, dict(alen=5000), upto=1); print(f"reference S1 dumps in {time.time() - t:.0f} s", flush=True)
    Z = np.load(npz)
    g = capi.Aligner(idx, alen=5000)
    def gpu_seeds(q, mode):
        g.set_option("seed_mode", mode); g.set_query(q); g.run_to(1); return g.seeds()
    def as_set(q, l, r): return set(zip(q.tolist(), l.tolist(), r.tolist()))
    for ci, (name, q) in enumerate(cases):
        want = as_set(Z[f"c{ci}_s1_qpos"], Z[f"c{ci}_s1_qlen"], Z[f"c{ci}_s1_rpos"])
        for mode in (1, 0, 2):
            got = as_set(*gpu_seeds(q, mode))
            miss = sorted(want - got); extra = sorted(got - want)
            print(f"{name} mode {mode}: ref {len(want)} gpu {len(got)} missing {len(miss)} extra {len(extra)}", flush=True)
            if mode == 1 and (miss or extra):
                wq = {}
                for s in want: wq.setdefault(s[0], []).append(s)
                gq = {}
                for s in got: gq.setdefault(s[0], []).append(s)
                qs_bad = sorted({s[0] for s in miss} | {s[0] for s in extra})[:12]
                st = np.array(qs_bad, np.int32); en = np.array([(x // 10000 + 1) * 10000 if (x // 10000 + 1) * 10000 < q.size else q.size for x in qs_bad], np.int32)
                g.set_query(q)
                ln, fr, loc = g.bwt_search_batch(st, en)
                for k, x in enumerate(qs_bad):
                    w = sorted(wq.get(x, [])); gg = sorted(gq.get(x, []))
                    print(f"   q={x} chunk_off={x % 10000}: ref len={w[0][1] if w else None} freq={len(w)} rpos={[s[2] for s in w][:6]} | gpu len={gg[0][1] if gg else None} freq={len(gg)} rpos={[s[2] for s in gg][:6]} | leaf op len={ln[k]} freq={fr[k]} loc={sorted(loc[k][:max(fr[k],0)].tolist())[:6]} | bases {bytes(q[x:x+24]).decode()}")
                # where do the missing seeds' reference positions lie
                mr = np.array([s[2] for s in miss], np.int64)
                if mr.size: print(f"   missing rpos: min {mr.min()} max {mr.max()}  >=2^32: {(mr >= 2**32).sum()}  >=G: {(mr >= G).sum()}  lens min {min(s[1] for s in miss)} max {max(s[1] for s in miss)}")
    g.close()
    print("DIAG DONE")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
