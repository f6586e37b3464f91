"""Live cross-check of the restatement against the real reference objects
(oracle/_ref/libgsref.so).  Skipped where oracle/_ref is absent."""
import os

import numpy as np
import pytest

from conftest import assert_stage_equal
from gsalign_amd import indexio, synth


@pytest.fixture(scope="module")
def op(oracle_built):
    if not oracle_built.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return oracle_built


@pytest.mark.parametrize("seed,params", [(31, {}), (32, dict(sen=1, clr=50)), (33, dict(one=1, ind=40, clr=300, alen=1000)), (34, dict(idy=95, slen=12))])
def test_complex_pairs_live(op, tmp_path, seed, params):
    refs, qrys = synth.make_complex(seed)
    rf, qf, px = str(tmp_path / "r.fa"), str(tmp_path / "q.fa"), str(tmp_path / "r")
    synth.write_fasta(rf, refs)
    op.ref_build_index(rf, px)
    o = op.Oracle(indexio.load_index(px), params)
    # A contig with no seed at all makes the reference read SeedVec[0] of an empty
    # vector (GSAlign.cpp:140,387) and segfault at -t 1; keep those out of the live run.
    keep = []
    for name, seq in qrys:
        o.set_query(seq); o.run_to(1)
        if o._call("seed_count") > 0:
            keep.append((name, seq))
    qrys = keep
    synth.write_fasta(qf, qrys)
    op.ref_dump_subprocess(px, qf, str(tmp_path / "ref.npz"), params)
    want = np.load(str(tmp_path / "ref.npz"))
    for ci, (name, seq) in enumerate(qrys):
        o.set_query(seq)
        assert_stage_equal(o.dump_stages(8), want, prefix=f"c{ci}_")
    o.close()


def test_low_divergence_long_dp_live(op, tmp_path):
    # 0.1 % divergence: few seeds, multi-kb DP problems
    refs, qrys = synth.make_pair(1500000, 1, 0.001, seed=3)
    rf, qf, px = str(tmp_path / "r.fa"), str(tmp_path / "q.fa"), str(tmp_path / "r")
    synth.write_fasta(rf, refs); synth.write_fasta(qf, qrys)
    op.ref_build_index(rf, px)
    op.ref_dump_subprocess(px, qf, str(tmp_path / "ref.npz"), {})
    want = np.load(str(tmp_path / "ref.npz"))
    o = op.Oracle(indexio.load_index(px))
    o.set_query(qrys[0][1])
    assert_stage_equal(o.dump_stages(8), want, prefix="c0_")
    o.close()


def test_adversarial_repeats_live(op, tmp_path):
    """repeat families with a copy-number spectrum (thousands of 1 %-divergent copies: `freq > MaxSeedFreq`
    reject-and-restart on every start inside a copy, bwt_search.cpp:177-182), microsatellites, N runs and soft-masked blocks
    (csrc/host/synth.cpp: gsah_c_synth_adversarial).  The restatement against the real reference objects after all 8 stages."""
    refs, qrys = synth.make_adversarial_pair(1_600_000, 2, 0.02, seed=91, n_run=60_000)
    qrys[1] = (qrys[1][0], synth.revcomp(qrys[1][1]))
    rf, qf, px = str(tmp_path / "r.fa"), str(tmp_path / "q.fa"), str(tmp_path / "r")
    synth.write_fasta(rf, refs); synth.write_fasta(qf, qrys)
    op.ref_build_index(rf, px)
    op.ref_dump_subprocess(px, qf, str(tmp_path / "ref.npz"), {})
    want = np.load(str(tmp_path / "ref.npz"))
    o = op.Oracle(indexio.load_index(px))
    for ci, (name, seq) in enumerate(qrys):
        o.set_query(seq)
        assert_stage_equal(o.dump_stages(8), want, prefix=f"c{ci}_")
    o.close()


def test_human_like_repeats_live(op, tmp_path):
    """Round 5: the human-like repeat spectrum (csrc/host/synth.cpp: gsah_c_synth_human_like -- Alu-, L1-, LTR-like families by age class, ancient
    repeats, segmental duplications, microsatellites, N runs, soft-masked blocks over ~45 % of the sequence).  The restatement against the real
    reference objects after all 8 stages, forward and reverse strand."""
    refs, qrys = synth.make_human_like_pair(2_000_000, 2, 0.015, seed=93, n_run=60_000)
    qrys[1] = (qrys[1][0], synth.revcomp(qrys[1][1]))
    rf, qf, px = str(tmp_path / "r.fa"), str(tmp_path / "q.fa"), str(tmp_path / "r")
    synth.write_fasta(rf, refs); synth.write_fasta(qf, qrys)
    op.ref_build_index(rf, px)
    op.ref_dump_subprocess(px, qf, str(tmp_path / "ref.npz"), {})
    want = np.load(str(tmp_path / "ref.npz"))
    o = op.Oracle(indexio.load_index(px))
    for ci, (name, seq) in enumerate(qrys):
        o.set_query(seq)
        assert_stage_equal(o.dump_stages(8), want, prefix=f"c{ci}_")
    o.close()


def test_ksw2_edge_shapes_live(op):
    """The pairs the striped GPU kernel is checked on (tools/dp_fuzz.py: query lengths around multiples of 64 / 128, one-row and
    1500-row reference sides, N bases, long pairs up to 5000 x 5000), here the restatement against the reference's own
    ksw2_alignment (ksw2_alignment.cpp:251-273) -- so the GPU test's checker is pinned on exactly those shapes."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import dp_fuzz
    ref = op.RefLib(None)
    s1, s2 = dp_fuzz.make_pairs(800, 5)
    l1, l2 = dp_fuzz.make_large_pairs(8, 5)
    for a, b in zip(s1 + l1, s2 + l2):
        assert op.oracle_ksw2(a, b) == ref.ksw2(a, b), (len(a), len(b))
