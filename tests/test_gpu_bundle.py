"""Several contigs in ONE pass (gsa_align_bundle; what gsa_align_many does with short contigs): every contig's result must be
exactly what it gets alone -- compared with the oracle, which aligns contig by contig as the reference does (GSAlign.cpp:483-548:
all per-sequence state is cleared between query sequences), and with gsa_align_contig of the same context field by field
(blocks, 16-byte records, both string pools)."""
import os

import numpy as np
import pytest

from gsalign_amd import capi, indexio, synth

pytestmark = pytest.mark.gpu


def _build(tmp_path, refs):
    from gsalign_amd import hostlib
    rf, px = str(tmp_path / "r.fa"), str(tmp_path / "r")
    synth.write_fasta(rf, refs); hostlib.build_index(rf, px)
    return indexio.load_index(px)


def _same_result(a, b, what):
    """blocks and records byte for byte (offsets included), the strings the records address, and the whole pools: they give a DP gap
    the room it can need at most -- m + n -- and the bytes no record owns are zero since late round 3 (they used to be whatever the
    buffer held: tools/stress_consistency.py saw bundles differ from call to call in exactly those bytes)"""
    for k in ("blocks", "frags"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (what, k)
    assert a["aln1"].size == b["aln1"].size and a["aln2"].size == b["aln2"].size, (what, "pool size")
    assert np.array_equal(a["aln1"], b["aln1"]) and np.array_equal(a["aln2"], b["aln2"]), (what, "pool bytes outside the records' strings")
    da, db = capi.result_as_dump(a, with_aln=True), capi.result_as_dump(b, with_aln=True)
    for k in ("aln1", "aln2"):
        assert np.array_equal(da[k], db[k]), (what, k)


def _check_bundle(g, o, seqs, names, oracle_on=None):
    """bundle == alone (every byte), and == the oracle for the contigs in oracle_on (default: all)."""
    got = g.align_bundle(seqs)
    assert len(got) == len(seqs)
    for k, seq in enumerate(seqs):
        alone = g.align_contig(seq)
        _same_result(got[k], alone, names[k])
        if o is not None and (oracle_on is None or k in oracle_on):
            o.set_query(seq); o.run_to(8); want = o.blocks(with_aln=True)
            d = capi.result_as_dump(got[k], with_aln=True)
            for key, v in want.items():
                assert np.array_equal(d[key], v), (names[k], key)
    return got


@pytest.mark.parametrize("seed,params", [(61, {}), (62, dict(sen=1, clr=50)), (63, dict(ind=40, clr=300)), (64, dict(one=1, idy=95))])
def test_bundle_vs_oracle_mixed_contigs(oracle_built, tmp_path, seed, params):
    """Contigs of every awkward shape in one bundle: lengths around the 10 000-bp chunk edge (no padding / one base of padding),
    shorter than a seed, empty, all N, reverse strand, two copies of the same contig (identical PosDiff values in two strides),
    a contig that matches nothing.  -ind 40 takes the PosDiff-sort path (no bitmap), -sen the dense search on every chunk."""
    refs, qrys = synth.make_pair_fast(1500000, 5, 0.02, seed=seed)
    idx = _build(tmp_path, refs)
    rng = np.random.default_rng(seed)
    r0 = refs[0][1]
    seqs = [q for _, q in qrys]
    seqs[1] = synth.revcomp(seqs[1])
    seqs += [seqs[0][:20000].copy(), seqs[0][:20001].copy(), seqs[2][:9999].copy(), np.frombuffer(b"ACGTACGTAC", np.uint8).copy(), np.zeros(0, np.uint8),
             np.full(12345, ord("N"), np.uint8), synth.random_genome(30000, rng), seqs[3].copy(), r0[1000:31000].copy()]
    names = [f"c{k}" for k in range(len(seqs))]
    o = oracle_built.Oracle(idx, params); g = capi.Aligner(idx, **params)
    got = _check_bundle(g, o, seqs, names)
    assert sum(r["blocks"].size for r in got) >= 5
    # the same contigs in another order and another split: results do not depend on the company a contig keeps
    perm = rng.permutation(len(seqs))
    again = g.align_bundle([seqs[i] for i in perm[:7]])
    for j, i in enumerate(perm[:7]):
        _same_result(again[j], got[i], f"perm {i}")
    o.close(); g.close()


def test_bundle_device_resident_and_wide_layout(oracle_built, tmp_path):
    """Device-resident contigs (one gather kernel builds the concatenation) under the >= 2^32-row index layout."""
    refs, qrys = synth.make_pair_fast(900000, 4, 0.03, seed=71)
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx); g = capi.Aligner(idx, wide=True)
    seqs = [q for _, q in qrys] + [qrys[0][1][:10000].copy()]
    host = _check_bundle(g, o, seqs, [f"c{k}" for k in range(len(seqs))])
    devs = [capi.DeviceContig(g.lib, 0, s) for s in seqs]
    dev = g.align_bundle(devs)
    for k in range(len(seqs)):
        _same_result(dev[k], host[k], f"device {k}")
    for d in devs:
        d.free()
    o.close(); g.close()


def test_bundle_yeast_sized_sen(tmp_path):
    """BASELINE configs[2] (16 contigs, 12 Mb, 2 %, -sen) as ONE bundle against the same contigs one by one (which
    test_config3_yeast_sized_sen holds against the oracle)."""
    from test_gpu_parity import YEAST_KB
    params = dict(sen=1, clr=50)
    refs, qrys = synth.make_pair_fast(0, 16, 0.02, seed=52, lengths=[1000 * k for k in YEAST_KB])
    qrys[3] = (qrys[3][0], synth.revcomp(qrys[3][1]))
    idx = _build(tmp_path, refs)
    g = capi.Aligner(idx, **params)
    _check_bundle(g, None, [q for _, q in qrys], [n for n, _ in qrys])
    g.close()


def test_align_many_bundles_short_contigs(oracle_built, tmp_path, monkeypatch):
    """gsa_align_many with and without bundles (GSA_MANY_NO_BUNDLE), three contexts, twelve contigs of 0.2 - 0.6 Mb and one of
    3 Mb that stays alone under GSA_BUNDLE_CONTIG = 1 Mb: the same bytes per contig either way, and the oracle's blocks."""
    monkeypatch.setenv("GSA_BUNDLE_CONTIG", "1000000")
    lens = [200000 + 37000 * k for k in range(12)] + [3000000]
    refs, qrys = synth.make_pair_fast(0, len(lens), 0.02, seed=72, lengths=lens)
    idx = _build(tmp_path, refs)
    g0 = capi.Aligner(idx); ctxs = [g0, g0.clone(), g0.clone()]
    seqs = [q for _, q in qrys]

    def run(bundle):
        out = {}

        def on_result(ci, res):
            out[ci] = g0._result(res)
            return 0
        capi.align_many(ctxs, seqs, on_result, bundle=bundle)
        return out
    a, b = run(True), run(False)
    assert sorted(a) == sorted(b) == list(range(len(seqs)))
    for ci in a:
        _same_result(a[ci], b[ci], f"contig {ci}")
    o = oracle_built.Oracle(idx)
    for ci in (0, 5, 11):
        o.set_query(seqs[ci]); o.run_to(8); want = o.blocks(with_aln=True)
        d = capi.result_as_dump(a[ci], with_aln=True)
        for key, v in want.items():
            assert np.array_equal(d[key], v), (ci, key)
    o.close()
    for g in ctxs[1:]:
        g.close()
    g0.close()


def test_prefetch_hides_upload_same_bytes(oracle_built, tmp_path, monkeypatch):
    """Round 4: a context uploads its NEXT contig (or bundle) into its second query slot while it aligns the current one
    (gsa_prefetch_contig / gsa_prefetch_bundle; gsa_align_many does it by itself).  (1) gsa_align_many with and without the prefetch
    (GSA_MANY_NO_PREFETCH), two contexts, bundles and single contigs mixed: the same bytes per contig.  (2) the calls by hand on one
    context -- prefetch(next); align(current) -- against plain gsa_align_contig and the oracle; state rules: a third waiting contig is
    refused, gsa_rewind after a prefetch took the previous contig's slot is refused, gsa_cancel_prefetch frees the slots."""
    monkeypatch.setenv("GSA_BUNDLE_CONTIG", "1000000")
    lens = [150000 + 41000 * k for k in range(9)] + [2500000, 1800000, 9999, 0, 1200000]
    refs, qrys = synth.make_pair_fast(0, len(lens), 0.02, seed=81, lengths=lens)
    idx = _build(tmp_path, refs)
    g0 = capi.Aligner(idx); ctxs = [g0, g0.clone()]
    seqs = [g0.pinned_copy(q) for _, q in qrys]

    def run(prefetch):
        out = {}

        def on_result(ci, res):
            out[ci] = g0._result(res)
            return 0
        capi.align_many(ctxs, seqs, on_result, prefetch=prefetch)
        return out
    a, b = run(True), run(False)
    assert sorted(a) == sorted(b) == list(range(len(seqs)))
    for ci in a:
        _same_result(a[ci], b[ci], f"contig {ci}")
    # (2) by hand
    g = ctxs[1]
    order = [9, 0, 10, 13, 3]
    plain = {ci: g.align_contig(seqs[ci]) for ci in order}
    g.prefetch_contig(seqs[order[0]])
    got = {}
    for k, ci in enumerate(order):
        if k + 1 < len(order):
            g.prefetch_contig(seqs[order[k + 1]])
        got[ci] = g.align_contig(seqs[ci])
    for ci in order:
        _same_result(got[ci], plain[ci], f"prefetched contig {ci}")
    o = oracle_built.Oracle(idx)
    for ci in (9, 3):
        o.set_query(np.array(seqs[ci])); o.run_to(8); want = o.blocks(with_aln=True)
        d = capi.result_as_dump(got[ci], with_aln=True)
        for key, v in want.items():
            assert np.array_equal(d[key], v), (ci, key)
    o.close()
    # state rules
    g.align_contig(seqs[0])
    g.prefetch_contig(seqs[1]); g.prefetch_contig(seqs[2])
    with pytest.raises(capi.GsaError):
        g.prefetch_contig(seqs[3])                  # two contigs wait already
    with pytest.raises(capi.GsaError):
        g.rewind()                                  # the second prefetch took contig 0's slot
    g.cancel_prefetch()
    r = g.align_contig(seqs[3]); g.rewind(); g.run_to(8)
    _same_result(g.blocks(), r, "rewind after cancel")
    # a prefetched contig nobody aligns does not get in the way of other contigs
    g.prefetch_contig(seqs[10])
    _same_result(g.align_contig(seqs[9]), plain[9], "other contig while one waits")
    _same_result(g.align_contig(seqs[10]), plain[10], "the waiting contig, later")
    for x in ctxs[1:]:
        x.close()
    g0.close()
