"""Parity of the HIP path (through the C ABI) with the oracle, on a real MI355X.

Bar: bit-exact -- every seed, group, block, gap record, score and gapped string."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_stage_equal
from gsalign_amd import capi, indexio, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["narrow", "wide"])
def gpu(request, cx_index):
    """Every test on the committed index runs twice: with the layout the text length selects (32-bit dense SA, 16-byte
    k-mer entries) and with GSA_CREATE_WIDE, the layout of a text with >= 2^32 BWT rows (a full human index)."""
    a = capi.Aligner(cx_index, wide=(request.param == "wide"))
    yield a
    a.close()


@pytest.fixture(scope="module")
def ora(oracle_built, cx_index):
    o = oracle_built.Oracle(cx_index)
    yield o
    o.close()


def test_bwt_search_leaf_operator(gpu, ora, cx_queries):
    rng = np.random.default_rng(1)
    for ci in (0, 1, 4):
        seq = cx_queries[ci][1]
        gpu.set_query(seq); ora.set_query(seq)
        starts = rng.integers(0, seq.size - 1, size=400).astype(np.int32)
        starts = np.array([s for s in starts if seq[s] in b"ACGTacgt"], np.int32)
        stops = np.minimum((starts // 10000 + 1) * 10000, seq.size).astype(np.int32)
        ln, fr, loc = gpu.bwt_search_batch(starts, stops)
        for i, (s, e) in enumerate(zip(starts, stops)):
            olen, olocs = ora.bwt_search(int(s), int(e))
            assert ln[i] == olen and fr[i] == olocs.size and np.array_equal(loc[i, :fr[i]], olocs), (ci, s, e)


def test_ksw2_leaf_operator_golden(gpu):
    d = np.load(os.path.join(GOLDEN, "ksw2_pairs.npz"))
    n = d["s1_off"].size - 1
    s1 = [d["s1"][d["s1_off"][i]:d["s1_off"][i + 1]].tobytes() for i in range(n)]
    s2 = [d["s2"][d["s2_off"][i]:d["s2_off"][i + 1]].tobytes() for i in range(n)]
    ops = gpu.ksw2_batch(s1, s2)
    for i in range(n):
        a1 = d["a1"][d["a1_off"][i]:d["a1_off"][i + 1]].tobytes(); a2 = d["a2"][d["a2_off"][i]:d["a2_off"][i + 1]].tobytes()
        assert capi.apply_ops(s1[i], s2[i], ops[i]) == (a1, a2), f"pair {i} m={len(s1[i])} n={len(s2[i])}"


@pytest.mark.parametrize("seed", [1, 2])
def test_ksw2_leaf_operator_edge_shapes(oracle_built, cx_index, seed):
    """Random pairs around the striped kernel's edges (query lengths around multiples of 64 and 128: a wave takes two
    64-column stripes; one-row and very long reference sides; N bases; identical and unrelated pairs), each against the
    oracle's ksw2 (ksw2_alignment.cpp:74-95), called twice so that the second call reuses buffers and epochs."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), "tools"))
    import dp_fuzz
    s1, s2 = dp_fuzz.make_pairs(1200, seed)
    if seed == 2:      # + long pairs around the LDS limit of the four-wave layout (3968 reference bases) up to 5000 x 5000
        l1, l2 = dp_fuzz.make_large_pairs(16, seed); s1 += l1; s2 += l2
    a = capi.Aligner(cx_index)
    try:
        for rep in range(2):
            ops = a.ksw2_batch(s1, s2)
            for i in range(len(s1)):
                assert capi.apply_ops(s1[i], s2[i], ops[i]) == oracle_built.oracle_ksw2(s1[i], s2[i]), f"rep {rep} pair {i} m={len(s1[i])} n={len(s2[i])}"
    finally:
        a.close()


@pytest.mark.parametrize("lane", ["512", "64", "8192", "0"])
def test_ksw2_leaf_operator_small_classes(oracle_built, golden_dir, lane):
    """The classes below the striped kernel: 3000 pairs over every corner of n <= 64, m + n - 1 <= 128 (tools/dp_fuzz.py,
    make_small_pairs) against the oracle's ksw2 (ksw2_alignment.cpp:74-95) -- one alignment per lane up to GSA_DP_LANE cells
    (k_dp_lane; default 512), one per wavefront above (k_dp_small); 64 and 8192 move the border to both ends of the domain, 0 is
    round 2's split (four per wavefront / one per wavefront).  Twice: the second call reuses the arena and the sorted tiles."""
    import json
    import subprocess
    import sys
    # (the switch is read once per process: a child process per setting)
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import dp_fuzz; from gsalign_amd import capi, indexio; from oracle import oracle_py\n"
            "idx = indexio.load_index(%r); a = capi.Aligner(idx); s1, s2 = dp_fuzz.make_small_pairs(3000, 5); bad = []\n"
            "for rep in range(2):\n"
            "    ops = a.ksw2_batch(s1, s2)\n"
            "    bad += [(rep, i, len(s1[i]), len(s2[i])) for i in range(len(s1)) if capi.apply_ops(s1[i], s2[i], ops[i]) != oracle_py.oracle_ksw2(s1[i], s2[i])]\n"
            "a.close(); print(json.dumps(bad[:10]))\n") % (os.path.dirname(os.path.dirname(GOLDEN)), os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), "tools"), os.path.join(golden_dir, "cx"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, GSA_DP_LANE=lane), timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1]) == [], (lane, r.stdout[-500:])


def test_gap_similarity_leaf_operator_golden(gpu, cx_queries):
    rows = np.load(os.path.join(GOLDEN, "gapsim.npz"))["rows"]
    for ci in np.unique(rows[:, 0]):
        sel = rows[rows[:, 0] == ci]
        gpu.set_query(cx_queries[int(ci)][1])
        got = gpu.gap_similarity_batch(sel[:, 1], sel[:, 2], sel[:, 3], sel[:, 4])
        assert np.array_equal(got, sel[:, 5].astype(np.int32)), np.flatnonzero(got != sel[:, 5])[:10]


@pytest.mark.parametrize("golden,params", [("cx_stages.npz", {}), ("cx_sen_stages.npz", dict(sen=1, clr=50))])
def test_stages_vs_golden(gpu, cx_queries, golden, params):
    want = np.load(os.path.join(GOLDEN, golden))
    gpu.set_params(**params)
    for ci, (name, seq) in enumerate(cx_queries):
        gpu.set_query(seq)
        assert_stage_equal(gpu.dump_stages(8), want, prefix=f"c{ci}_")
    gpu.set_params()


@pytest.mark.parametrize("seed,params,wide", [(41, {}, False), (42, dict(sen=1, clr=50), False), (43, dict(one=1, ind=40, clr=300, alen=1000), False), (44, dict(idy=95, slen=12), False),
                                              (45, {}, True), (46, dict(sen=1, clr=50), True)])
def test_stages_vs_oracle_fresh_inputs(oracle_built, tmp_path, seed, params, wide):
    # a fresh complex pair; index built by the reference's bwt_index when oracle/_ref travelled, else skip
    if not oracle_built.have_ref():
        pytest.skip("oracle/_ref not present")
    refs, qrys = synth.make_complex(seed)
    rf, px = str(tmp_path / "r.fa"), str(tmp_path / "r")
    synth.write_fasta(rf, refs); oracle_built.ref_build_index(rf, px)
    idx = indexio.load_index(px)
    o = oracle_built.Oracle(idx, params); g = capi.Aligner(idx, wide=wide, **params)
    for name, seq in qrys:
        o.set_query(seq); g.set_query(seq)
        want = o.dump_stages(8)
        assert_stage_equal(g.dump_stages(8), want)
    o.close(); g.close()


def test_align_contig_drop_in(gpu, ora, cx_queries):
    gpu.set_profiling(False, count_blocks=True)      # accounting build: exact algorithmic Occ-block count
    for name, seq in cx_queries[:5]:
        ora.set_query(seq); ora.run_to(8)
        want = ora.blocks(with_aln=True)
        r = gpu.align_contig(seq)
        got = gpu.blocks_as_dump(with_aln=True)
        for k, v in want.items():
            assert np.array_equal(got[k], v), (name, k)
        c = gpu.counters(); oc = ora.counters()
        assert c[2] == oc[2] and c[3] == oc[3] and c[0] == oc[0], (c, oc)      # hits, seeds, algorithmic Occ blocks
        assert c[7] >= c[0]
    gpu.set_profiling(False)
    for name, seq in cx_queries[:3]:                 # default build: same result
        ora.set_query(seq); ora.run_to(8); want = ora.blocks(with_aln=True)
        gpu.align_contig(seq); got = gpu.blocks_as_dump(with_aln=True)
        for k, v in want.items():
            assert np.array_equal(got[k], v), (name, k)


def test_midsize_pair_2pct(oracle_built, tmp_path):
    if not oracle_built.have_ref():
        pytest.skip("oracle/_ref not present")
    refs, qrys = synth.make_pair(1000000, 2, 0.02, seed=9)
    rf, px = str(tmp_path / "r.fa"), str(tmp_path / "r")
    synth.write_fasta(rf, refs); oracle_built.ref_build_index(rf, px)
    idx = indexio.load_index(px)
    o = oracle_built.Oracle(idx); g = capi.Aligner(idx)
    for name, seq in qrys:
        o.set_query(seq); g.set_query(seq)
        assert_stage_equal(g.dump_stages(8), o.dump_stages(8))
    o.close(); g.close()


@pytest.mark.parametrize("total,ncontig,div,seed,params", [
    (1500000, 1, 0.001, 3, {}),                       # 0.1 %: few seeds, multi-kb DP problems (striped kernel, traceback tiles)
    (2000000, 3, 0.01, 12, {}),                       # 1 %
    (600000, 2, 0.02, 13, dict(sen=1, clr=50)),       # -sen: 5-bp stride, many tiny groups, candidate-buffer growth
    (800000, 2, 0.08, 14, dict(slen=12, idy=60)),     # high divergence, short seeds
    (12000000, 1, 0.02, 21, {}),                      # 12 Mb contig: window chain in global memory, > 2048 striped DP jobs, multi-tile scans
    (50000000, 1, 0.02, 22, {}),                      # 50 Mb contig: two size classes of striped jobs, grid-wide window chain, > 4096 early gaps
    (50000000, 1, 0.02, 23, dict(sen=1, clr=50)),     # round 6: -sen at 50 Mb (BASELINE configs[2]'s mode at 4x its size): 5 000 chunks through the sweep, ~13 M seeds, the PosDiff byte map
    (3000000, 2, 0.05, 15, {}),                       # 5 %: short seeds, dense gaps, many small DP jobs
    (1000000, 1, 0.02, 16, dict(ind=40)),             # MaxIndelSize > 31: grouping by the PosDiff sort instead of the bitmap
    (1500000, 1, 0.003, 17, dict(sen=1, clr=50)),     # -sen at low divergence: long seeds cut every 5 bases, multi-kb gaps
    (3000000, 1, 0.01, 18, dict(alen=5000)),          # -alen 5000 (BASELINE configs[4]): only long blocks survive AddAlnBlock / the re-split rules
    (2000000, 3, 0.01, 12, dict(wide=True)),          # the >= 2^32-row layout (GSA_CREATE_WIDE) on the cases above ...
    (600000, 2, 0.02, 13, dict(sen=1, clr=50, wide=True)),
    (12000000, 1, 0.02, 21, dict(wide=True)),         # ... including the grid-wide paths of a 12 Mb contig
    (3000000, 1, 0.01, 18, dict(alen=5000, wide=True)),
])
def test_scaled_pairs_vs_oracle(oracle_built, tmp_path, total, ncontig, div, seed, params):
    """Larger synthetic pairs than the committed fixtures; index from OUR builder, result vs the oracle."""
    from gsalign_amd import hostlib
    params = dict(params); wide = params.pop("wide", False)
    refs, qrys = synth.make_pair(total, ncontig, div, seed=seed)
    if ncontig > 1:
        qrys[-1] = (qrys[-1][0], synth.revcomp(qrys[-1][1]))
    rf, px = str(tmp_path / "r.fa"), str(tmp_path / "r")
    synth.write_fasta(rf, refs); hostlib.build_index(rf, px)
    idx = indexio.load_index(px)
    o = oracle_built.Oracle(idx, params); g = capi.Aligner(idx, wide=wide, **params)
    for name, seq in qrys:
        o.set_query(seq); o.run_to(8); want = o.blocks(with_aln=True)
        g.align_contig(seq); got = g.blocks_as_dump(with_aln=True)
        for k, v in want.items():
            assert np.array_equal(got[k], v), (name, k)
        assert want["b_score"].size > 0
    o.close(); g.close()


# S. cerevisiae S288C chromosome lengths I..XVI (kb, rounded): BASELINE configs[2] is this genome against a 2 %-divergent copy, -sen
YEAST_KB = [230, 813, 317, 1532, 577, 270, 1091, 563, 440, 746, 667, 1078, 924, 784, 1091, 948]


def _build(tmp_path, refs):
    from gsalign_amd import hostlib
    rf, px = str(tmp_path / "r.fa"), str(tmp_path / "r")
    synth.write_fasta(rf, refs); hostlib.build_index(rf, px)
    return indexio.load_index(px)


def _same_as_oracle(o, g, qrys):
    for name, seq in qrys:
        o.set_query(seq); o.run_to(8); want = o.blocks(with_aln=True)
        g.align_contig(seq); got = g.blocks_as_dump(with_aln=True)
        for k, v in want.items():
            assert np.array_equal(got[k], v), (name, k)


def test_config3_yeast_sized_sen(oracle_built, tmp_path):
    """BASELINE configs[2] as SURVEY 8(d) specifies it: 16 contigs totalling 12 Mb, 2 % divergence, -sen (forces -slen 10,
    5-bp seed stride, -clr 50): ~3 M seeds, ~10^5 seed groups per Mb.  Every block, record and gapped string vs the oracle."""
    params = dict(sen=1, clr=50)
    refs, qrys = synth.make_pair_fast(0, 16, 0.02, seed=52, lengths=[1000 * k for k in YEAST_KB])
    qrys[3] = (qrys[3][0], synth.revcomp(qrys[3][1]))
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx, params); g = capi.Aligner(idx, **params)
    _same_as_oracle(o, g, qrys)
    assert int(g.counters()[3]) > 100000
    o.close(); g.close()


@pytest.mark.parametrize("total,div,seed,params", [
    (12000000, 0.02, 51, {}),                          # SURVEY 8(d) repeat-stress variant at yeast size, defaults
    (2000000, 0.01, 53, dict(sen=1, clr=50)),          # the same under -sen (start+5 stride through the repeats)
    (3000000, 0.02, 54, dict(wide=True)),              # and in the >= 2^32-row layout
])
def test_repeat_stress_vs_oracle(oracle_built, tmp_path, total, div, seed, params):
    """A 300-bp family (copies 10 % divergent) covering 10 % of the genome + a 150-copy tandem array of a 40-bp unit:
    searches that end with more than MaxSeedFreq hits and restart one base further (bwt_search.cpp:177-182,
    GSAlign.cpp:91), multi-hit seeds (up to 100 located hits each), query positions with several hits per group."""
    params = dict(params); wide = params.pop("wide", False)
    refs, qrys = synth.make_pair_fast(total, 1, div, seed=seed, repeats=True)
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx, params); g = capi.Aligner(idx, wide=wide, **params)
    _same_as_oracle(o, g, qrys)
    c = g.counters()
    assert int(c[2]) > 0 and int(c[2]) == int(o.counters()[2])      # same number of located hits
    o.close(); g.close()


@pytest.mark.parametrize("total,ncontig,div,seed,params", [
    (12000000, 3, 0.02, 71, {}),                       # 12 Mb, every injection, defaults
    (3000000, 1, 0.01, 72, dict(sen=1, clr=50)),       # under -sen
    (4000000, 2, 0.03, 73, dict(wide=True)),           # in the >= 2^32-row layout
])
def test_adversarial_repeats_vs_oracle(oracle_built, tmp_path, total, ncontig, div, seed, params):
    """eight repeat families with a copy-number spectrum (1-15 % divergent copies, up to thousands of them:
    the `freq > MaxSeedFreq` reject-and-restart path of bwt_search.cpp:177-182 on a large fraction of the starts), microsatellites,
    two long N runs and soft-masked blocks -- every block, record and gapped string vs the oracle (itself pinned on such input
    against the live reference: test_oracle_vs_reference.py::test_adversarial_repeats_live).  Also reports how the seed search
    went: chunks redone by the dense search."""
    params = dict(params); wide = params.pop("wide", False)
    refs, qrys = synth.make_adversarial_pair(total, ncontig, div, seed=seed, n_run=300000)
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx, params); g = capi.Aligner(idx, wide=wide, **params)
    dense = 0
    for name, seq in qrys:
        o.set_query(seq); o.run_to(8); want = o.blocks(with_aln=True)
        g.align_contig(seq); got = g.blocks_as_dump(with_aln=True); dense += int(g.seed_stats()[1])
        for k, v in want.items():
            assert np.array_equal(got[k], v), (name, k)
    print(f"adversarial {total} bp {params}: {dense} chunks of {sum((q.size + 9999) // 10000 for _, q in qrys)} took the dense search")
    c = g.counters()
    assert int(c[2]) > 0 and int(c[2]) == int(o.counters()[2])
    o.close(); g.close()


@pytest.mark.parametrize("total,ncontig,div,seed,params", [(10000000, 2, 0.01, 81, {}), (3000000, 1, 0.02, 82, dict(wide=True))])
def test_human_like_repeats_vs_oracle(oracle_built, tmp_path, total, ncontig, div, seed, params):
    """Round 5: the interspersed-repeat spectrum of a primate genome over ~45 % of the sequence -- an Alu-like family in
    three age classes, 5'-truncated L1-like copies, LTR-like families with solo LTRs, ancient repeats, 20-kb segmental duplications at 1-3 %,
    microsatellites, N runs, soft-masked blocks (csrc/host/synth.cpp: gsah_c_synth_human_like; bench.py --workload human_like) -- every block,
    record and gapped string vs the oracle (pinned on such input against the live reference: test_oracle_vs_reference.py::test_human_like_repeats_live)."""
    params = dict(params); wide = params.pop("wide", False)
    refs, qrys = synth.make_human_like_pair(total, ncontig, div, seed=seed, n_run=200000)
    qrys[-1] = (qrys[-1][0], synth.revcomp(qrys[-1][1]))
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx, params); g = capi.Aligner(idx, wide=wide, **params)
    _same_as_oracle(o, g, qrys)
    o.close(); g.close()


@pytest.mark.parametrize("shape,params", [(0, {}), (1, {}), (0, dict(sen=1, clr=50)), (1, dict(sen=1, clr=50, wide=True))])
def test_sweep_launch_shapes_vs_oracle(oracle_built, tmp_path, shape, params):
    """k_dense_sweep's two launch shapes (round 4: four chunks per workgroup with the 160-start segments drawn by the lanes from an LDS counter;
    one chunk per four-wave workgroup with 40-start segments), each forced onto EVERY chunk of an adversarial pair (`seed_mode` 0, `sweep_shape`),
    with and without -sen: the shape is chosen by the number of dense chunks otherwise, and inputs of test size would only ever see one of them.
    Every block, record and gapped string vs the oracle; regime: bwt_search.cpp:177-182, GSAlign.cpp:75-91."""
    params = dict(params); wide = params.pop("wide", False)
    refs, qrys = synth.make_adversarial_pair(3000000, 2, 0.01, seed=74 + shape, n_run=100000)
    qrys[1] = (qrys[1][0], synth.revcomp(qrys[1][1]))
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx, params); g = capi.Aligner(idx, wide=wide, **params)
    g.set_option("seed_mode", 0); g.set_option("sweep_shape", shape)
    _same_as_oracle(o, g, qrys)
    assert int(g.seed_stats()[1]) == sum((q.size + 9999) // 10000 for _, q in qrys[-1:])      # every chunk of the last contig went the dense way
    c = g.counters()
    assert int(c[2]) > 0 and int(c[2]) == int(o.counters()[2])
    o.close(); g.close()


@pytest.mark.parametrize("k,wide", [(15, False), (15, True)])
def test_long_kmer_table(oracle_built, tmp_path, monkeypatch, k, wide):
    """Human-chromosome-sized texts get a k-mer jump table of k = 15 (16 GiB; 32 with wide entries) -- the table length follows
    the text length, so on a short text GSA_KMER_K forces it: same seeds, blocks and strings as
    the oracle, both entry widths."""
    monkeypatch.setenv("GSA_KMER_K", str(k))
    refs, qrys = synth.make_pair_fast(2000000, 2, 0.02, seed=60, repeats=True)
    qrys[1] = (qrys[1][0], synth.revcomp(qrys[1][1]))
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx); g = capi.Aligner(idx, wide=wide)
    _same_as_oracle(o, g, qrys)
    # round 5: with k = MinSeedLength the presence table is derived from the k-mer table (what a human index gets); the same table from the
    # scan of the text (the other build path: option pres_from_kmer 0, rebuilt by a parameter change and back) gives the same result
    g.set_option("pres_from_kmer", 0); g.set_params(slen=14); g.set_params()
    _same_as_oracle(o, g, qrys)
    o.close(); g.close()


@pytest.mark.parametrize("mode,params", [(2, dict(sen=1, clr=50)), (2, {}), (2, dict(sen=1, clr=50, wide=True)), (0, dict(sen=1, clr=50))])
def test_pd_byte_map_vs_oracle(oracle_built, tmp_path, mode, params):
    """Round 5: under -sen a chunk holds thousands of chance hits, each on a PosDiff-bitmap word of its own, and k_seed_select's device-scope atomics on
    those words were what `locate` cost (GSAlign.cpp:88 start += 5, :80-86 one seed per located hit).  Such contigs now mark a BYTE per PosDiff value
    with plain stores and k_pd_pack makes the bitmap of it (option pd_bytes: 1 = by the hit count, the default; here 2 = every contig, 0 = never).
    Every block, record and gapped string vs the oracle, contig after contig on one context (the byte map must be left clean), then the same contigs
    as bundles through gsa_align_many on two contexts."""
    params = dict(params); wide = params.pop("wide", False)
    refs, qrys = synth.make_pair_fast(2400000, 5, 0.02, seed=66, repeats=True)
    qrys[2] = (qrys[2][0], synth.revcomp(qrys[2][1]))
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx, params); g = capi.Aligner(idx, wide=wide, **params)
    g.set_option("pd_bytes", mode)
    _same_as_oracle(o, g, qrys)
    _same_as_oracle(o, g, qrys[:2])
    want = []
    for name, seq in qrys:
        o.set_query(seq); o.run_to(8); want.append(o.blocks(with_aln=False))
    g2 = g.clone(); g2.set_params(**params); g2.set_option("pd_bytes", mode)
    got = {}

    def on_result(ci, res):
        B = np.ctypeslib.as_array(capi.C.cast(res.blocks, capi.C.POINTER(capi.C.c_uint8)), shape=(res.n_blocks * 40,)).view(capi.BLOCK_DT).copy()
        got[ci] = (B["score"].copy(), B["aln_len"].copy(), int(res.n_frags))
        return 0

    capi.align_many([g, g2], [q for _, q in qrys] * 3, on_result)
    assert sorted(got) == list(range(3 * len(qrys)))
    for ci, (sc, al, nf) in got.items():
        w = want[ci % len(qrys)]
        assert np.array_equal(sc, w["b_score"]) and np.array_equal(al, w["b_aln_len"]) and nf == int(w["b_nfrag"].sum()), ci
    g2.close(); o.close(); g.close()


def test_host_register_in_place(oracle_built, tmp_path):
    """gsa_host_register (round 5): a sequence page-locked where the loader put it aligns to the same result as from pageable memory and as the
    oracle says; registering twice / unregistering something else reports GSA_ERR_HIP instead of failing later."""
    import ctypes as C
    refs, qrys = synth.make_pair_fast(1200000, 1, 0.02, seed=64)
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx); g = capi.Aligner(idx)
    q = np.ascontiguousarray(qrys[0][1])
    lib = g.lib
    lib.gsa_host_register.argtypes = [C.c_void_p, C.c_size_t]; lib.gsa_host_unregister.argtypes = [C.c_void_p]
    assert lib.gsa_host_register(C.c_void_p(q.ctypes.data), q.size) == 0
    _same_as_oracle(o, g, [("q", q)])
    assert lib.gsa_host_unregister(C.c_void_p(q.ctypes.data)) == 0
    _same_as_oracle(o, g, [("q", q)])
    assert lib.gsa_host_register(None, 10) != 0 and lib.gsa_host_unregister(C.c_void_p(q.ctypes.data)) != 0
    o.close(); g.close()


def test_striped_dp_fallback_path(oracle_built, tmp_path, monkeypatch):
    """The safety net behind the striped DP's bounded hand-off wait: one job per launch (GSA_DP_SAFE=1 forces it)."""
    monkeypatch.setenv("GSA_DP_SAFE", "1")
    refs, qrys = synth.make_pair_fast(600000, 1, 0.002, seed=57)
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx); g = capi.Aligner(idx)
    _same_as_oracle(o, g, qrys)
    assert int(g.counters()[5]) >= 0
    o.close(); g.close()


def test_align_many_contexts(oracle_built, tmp_path):
    """gsa_align_many: contigs handed to three contexts by the library's own worker threads; every result vs the oracle."""
    refs, qrys = synth.make_pair_fast(3000000, 6, 0.02, seed=58)
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx)
    want = []
    for name, seq in qrys:
        o.set_query(seq); o.run_to(8); want.append(o.blocks(with_aln=False))
    o.close()
    g0 = capi.Aligner(idx); ctxs = [g0, g0.clone(), g0.clone()]
    got = {}

    def on_result(ci, res):
        nb = res.n_blocks
        B = np.ctypeslib.as_array(capi.C.cast(res.blocks, capi.C.POINTER(capi.C.c_uint8)), shape=(nb * 40,)).view(capi.BLOCK_DT).copy()
        got[ci] = (B["score"].copy(), B["aln_len"].copy(), int(res.n_frags))
        return 0

    capi.align_many(ctxs, [q for _, q in qrys] * 2, on_result)
    assert sorted(got) == list(range(2 * len(qrys)))
    for ci, (sc, al, nf) in got.items():
        w = want[ci % len(qrys)]
        assert np.array_equal(sc, w["b_score"]) and np.array_equal(al, w["b_aln_len"]) and nf == int(w["b_nfrag"].sum()), ci
    for g in ctxs[1:]:
        g.close()
    g0.close()


@pytest.mark.parametrize("params", [{}, dict(sen=1, clr=50), dict(ind=40)])
def test_contig_seeded_in_chunk_ranges(oracle_built, tmp_path, params):
    """SURVEY 8(e) for one long contig: stage 1 on three chunk ranges by three contexts (what three GPUs would do), the hits
    of the other ranges imported by the owner -- once through host memory, once device to device -- which finishes the
    contig: identical to the oracle (and to gsa_align_contig on one context)."""
    refs, qrys = synth.make_pair_fast(1500000, 1, 0.02, seed=59, repeats=True)
    q = qrys[0][1]
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx, params)
    o.set_query(q); o.run_to(8); want = o.blocks(with_aln=True); o.close()
    own = capi.Aligner(idx, **params); a = own.clone(); b = own.clone()
    a.set_params(**params); b.set_params(**params)
    n_chunks = (q.size + 9999) // 10000
    cuts = [0, n_chunks // 3, n_chunks // 3 + 1, n_chunks + 5]          # (a one-chunk range and a range that overshoots the contig)
    n0 = own.seed_chunks(q, cuts[0], cuts[1]); n1 = a.seed_chunks(q, cuts[1], cuts[2]); n2 = b.seed_chunks(q, cuts[2], cuts[3])
    k1, v1 = a.export_hits(); own.import_hits(k1, v1)                    # through host memory (any transport)
    import ctypes
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")                    # (the runtime libgsa_hip.so itself uses)
    dk, dv = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dk), ctypes.c_size_t(8 * max(n2, 1))) == 0 and hip.hipMalloc(ctypes.byref(dv), ctypes.c_size_t(4 * max(n2, 1))) == 0
    b.export_hits(dk.value, dv.value)
    own.import_hits(dk.value, dv.value, n2)                              # device to device (what an RCCL recv buffer is)
    hip.hipFree(dk); hip.hipFree(dv)
    own.finish_contig()
    got = own.blocks_as_dump(with_aln=True)
    for k, v in want.items():
        assert np.array_equal(got[k], v), k
    assert n0 + n1 + n2 == int(own.counters()[3]) and min(n0, n2) > 0
    # the same context takes whole contigs again afterwards
    own.align_contig(q); got = own.blocks_as_dump(with_aln=True)
    for k, v in want.items():
        assert np.array_equal(got[k], v), k
    for g in (a, b, own):
        g.close()


def test_align_many_splits_one_contig_over_idle_contexts(oracle_built, tmp_path, monkeypatch):
    """gsa_align_many with fewer contigs than contexts: the contexts are grouped per contig and a long contig is seeded by chunk
    range on its whole group (gsa_seed_chunks / gsa_hit_buffers / gsa_import_hits / gsa_finish_contig inside the library, hits device
    to device) -- two contigs on five contexts, and one on three; every result vs the oracle."""
    monkeypatch.setenv("GSA_SPLIT_MIN", "200000")
    refs, qrys = synth.make_pair_fast(2400000, 2, 0.02, seed=63, repeats=True, lengths=[1700000, 700000])
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx)
    want = []
    for name, seq in qrys:
        o.set_query(seq); o.run_to(8); want.append(o.blocks(with_aln=True))
    o.close()
    g0 = capi.Aligner(idx); ctxs = [g0] + [g0.clone() for _ in range(4)]
    got = {}

    def on_result(ci, res):
        got[ci] = g_of_result(res); return 0

    def g_of_result(res):
        nb, nf, na = res.n_blocks, res.n_frags, res.n_aln
        B = np.ctypeslib.as_array(capi.C.cast(res.blocks, capi.C.POINTER(capi.C.c_uint8)), shape=(nb * 40,)).view(capi.BLOCK_DT).copy()
        F = capi.expand_recs(np.ctypeslib.as_array(capi.C.cast(res.recs, capi.C.POINTER(capi.C.c_uint8)), shape=(nf * 16,)).view(capi.REC_DT).copy())
        a1 = np.ctypeslib.as_array(capi.C.cast(res.aln1, capi.C.POINTER(capi.C.c_uint8)), shape=(na,)).copy()
        return B, F, a1

    for sel, nctx in (([0, 1], 5), ([0], 3), ([1], 2)):
        got.clear()
        capi.align_many(ctxs[:nctx], [qrys[i][1] for i in sel], on_result)
        for k, i in enumerate(sel):
            B, F, a1 = got[k]; w = want[i]
            idxs = np.concatenate([np.arange(o_, o_ + n_) for o_, n_ in zip(B["frag_off"], B["n_frag"])])
            Fo = F[idxs]
            assert np.array_equal(B["score"], w["b_score"]) and np.array_equal(B["aln_len"], w["b_aln_len"]), (sel, nctx)
            assert np.array_equal(Fo["qpos"], w["f_qpos"]) and np.array_equal(Fo["rpos"], w["f_rpos"]) and np.array_equal(Fo["aln_len"], w["f_alnlen"])
            segs = [a1[o_:o_ + n_] for o_, n_ in zip(Fo["aln_off"], Fo["aln_len"]) if n_]
            assert np.array_equal(np.concatenate(segs) if segs else np.zeros(0, np.uint8), w["aln1"])
    for g in ctxs[1:]:
        g.close()
    g0.close()


def test_device_resident_query_and_compact_records(gpu, ora, cx_queries):
    """gsa_align_contig_device: the contig already in device memory (gsa_device_alloc / gsa_device_upload), used in place -- same
    result as the host-buffer call; through gsa_align_many with GSA_MANY_DEVICE too.  The result's 16-byte records expand to
    the oracle's FragPair_t values (blocks_as_dump goes through capi.expand_recs)."""
    for ci in (0, 2, 6):
        seq = cx_queries[ci][1]
        d = gpu.device_copy(seq)
        gpu.align_contig_device(d)
        got = gpu.blocks_as_dump(with_aln=True)
        ora.set_query(seq); ora.run_to(8); want = ora.blocks(with_aln=True)
        for k, v in want.items():
            assert np.array_equal(got[k], v), (ci, k)
    seen = {}

    def on_result(ci, res):
        seen[ci] = (int(res.n_blocks), int(res.n_frags), int(res.n_aln)); return 0
    devs = [gpu.device_copy(q) for _, q in cx_queries[:4]]
    capi.align_many([gpu], devs, on_result)
    for ci in range(4):
        gpu.align_contig(cx_queries[ci][1]); r = gpu.raw_result()
        assert seen[ci] == (int(r.n_blocks), int(r.n_frags), int(r.n_aln)), ci
    # a host pointer is refused, and so is a misaligned device pointer
    res = capi.Result()
    host = np.ascontiguousarray(cx_queries[0][1])
    assert gpu.lib.gsa_align_contig_device(gpu.ctx, capi.C.c_void_p(host.ctypes.data), capi.C.c_int32(host.size), capi.C.byref(res)) == -1
    assert gpu.lib.gsa_align_contig_device(gpu.ctx, capi.C.c_void_p(devs[0].ptr + 4), capi.C.c_int32(1000), capi.C.byref(res)) == -1


def test_clone_parent_cannot_rebuild_shared_seed_tables(cx_index, cx_queries, ora):
    """a clone reads its parent's presence bitmap / short k-mer table through copied pointers, so the parent may not
    rebuild them while clones live (GSA_ERR_STATE); a clone changing its OWN parameters builds its own and stays exact."""
    g0 = capi.Aligner(cx_index); cl = g0.clone()
    with pytest.raises(capi.GsaError, match="cloned from this one"):
        g0.set_params(sen=1, clr=50)
    g0.set_params(idy=80); g0.set_params()                              # (parameters that do not touch the tables are fine)
    cl.set_params(sen=1, clr=50)                                        # the clone's own tables
    seq = cx_queries[3][1]
    o2 = type(ora)(cx_index, dict(sen=1, clr=50)); o2.set_query(seq); o2.run_to(8); want_sen = o2.blocks(with_aln=True); o2.close()
    ora.set_query(seq); ora.run_to(8); want = ora.blocks(with_aln=True)
    cl.align_contig(seq); g0.align_contig(seq)
    got_sen, got = cl.blocks_as_dump(with_aln=True), g0.blocks_as_dump(with_aln=True)
    for k, v in want_sen.items():
        assert np.array_equal(got_sen[k], v), k
    for k, v in want.items():
        assert np.array_equal(got[k], v), k
    cl.close()
    g0.set_params(sen=1, clr=50); g0.align_contig(seq)                  # no clone left: allowed again
    got_sen = g0.blocks_as_dump(with_aln=True)
    for k, v in want_sen.items():
        assert np.array_equal(got_sen[k], v), k
    g0.close()


def test_split_contig_retry_and_state_rules(oracle_built, tmp_path, monkeypatch):
    """gsa_finish_contig repeats stages 2-8 with one DP job per launch after a stripe hand-off time-out (forced once
    by the test hook GSA_DP_FAKE_TIMEOUT), from the hits it still holds; gsa_run_to is refused while a context holds a chunk range."""
    monkeypatch.setenv("GSA_DP_FAKE_TIMEOUT", "1")
    refs, qrys = synth.make_pair_fast(1200000, 1, 0.01, seed=61, repeats=True)
    q = qrys[0][1]
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx); o.set_query(q); o.run_to(8); want = o.blocks(with_aln=True); o.close()
    own = capi.Aligner(idx); other = own.clone()
    n_chunks = (q.size + 9999) // 10000
    own.seed_chunks(q, 0, n_chunks // 2); other.seed_chunks(q, n_chunks // 2, n_chunks)
    with pytest.raises(capi.GsaError, match="gsa_finish_contig"):
        own.run_to(8)
    k, v = other.export_hits(); own.import_hits(k, v)
    own.finish_contig()                                                  # first pass reports the (fake) time-out, the retry answers
    got = own.blocks_as_dump(with_aln=True)
    for key, val in want.items():
        assert np.array_equal(got[key], val), key
    own.align_contig(q)                                                  # hook used up: the plain path, same answer
    got = own.blocks_as_dump(with_aln=True)
    for key, val in want.items():
        assert np.array_equal(got[key], val), key
    other.close(); own.close()


def test_two_contexts_share_one_index(oracle_built, tmp_path):
    """gsa_clone: two contexts on one GPU, one device index, driven from two host threads on different contigs at the
    same time -- results identical to the oracle's (and so to a single context's)."""
    import threading
    refs, qrys = synth.make_pair_fast(4000000, 8, 0.02, seed=55)
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx)
    want = []
    for name, seq in qrys:
        o.set_query(seq); o.run_to(8); want.append(o.blocks(with_aln=True))
    o.close()
    g0 = capi.Aligner(idx); ctxs = [g0, g0.clone(), g0.clone()]
    errs = []

    def work(k):
        try:
            g = ctxs[k]
            for rep in range(3):
                for ci in range(k, len(qrys), len(ctxs)):
                    g.align_contig(g.pinned_copy(qrys[ci][1]) if rep == 1 else qrys[ci][1])
                    got = g.blocks_as_dump(with_aln=True)
                    for key, v in want[ci].items():
                        assert np.array_equal(got[key], v), (k, ci, key)
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(ctxs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for g in ctxs[1:]:
        g.close()
    g0.close()


@pytest.mark.parametrize("wide", [False, True])
def test_clone_to_device_copies_the_built_index(oracle_built, tmp_path, wide):
    """gsa_clone_to_device: a context whose device index is a device-to-device copy of the parent's finished tables (SURVEY 8(e)'s "build once,
    copy over xGMI"; the state being replicated is bwt_index.cpp:147-264's).  On a one-GPU box the target is the parent's own device: the copy is still a
    second, independent index -- it answers like the oracle AFTER the parent (and the clone it was taken from, which had built -sen tables of its
    own) are destroyed, and it can be gsa_clone'd itself."""
    refs, qrys = synth.make_pair_fast(1500000, 3, 0.02, seed=91)
    idx = _build(tmp_path, refs)
    want = {}
    for sen in (0, 1):
        o = oracle_built.Oracle(idx, dict(sen=sen, clr=50) if sen else {})
        want[sen] = []
        for name, seq in qrys:
            o.set_query(seq); o.run_to(8); want[sen].append(o.blocks(with_aln=True))
        o.close()
    parent = capi.Aligner(idx, wide=wide)
    child = parent.clone(); child.set_params(sen=1, clr=50)             # builds a presence table and a short k-mer table of its own
    copy_default = parent.clone_to_device(0)
    copy_sen = child.clone_to_device(0)                                  # must carry the CHILD's tables, not the index owner's
    child.close(); parent.close()                                        # the copies own everything they read
    grand = copy_default.clone()
    for g, sen in ((copy_default, 0), (copy_sen, 1), (grand, 0)):
        for ci, (name, seq) in enumerate(qrys):
            g.align_contig(seq)
            got = g.blocks_as_dump(with_aln=True)
            for key, v in want[sen][ci].items():
                assert np.array_equal(got[key], v), (sen, ci, key)
    grand.close(); copy_default.close(); copy_sen.close()


@pytest.mark.parametrize("wide", [False, True])
def test_create_from_pac_bytes(golden_dir, cx_index, cx_queries, ora, wide):
    """GSA_CREATE_REF_PAC: gsa_create given the bytes of the .pac file instead of RefSequence -- RestoreReferenceInfo's unpacking (bwt_index.cpp:229-264: forward
    strand + reverse complement) runs on the device -- answers exactly like a context created from the unpacked text (and so like the oracle)."""
    pac = np.fromfile(os.path.join(golden_dir, "cx.pac"), dtype=np.uint8)
    g = capi.Aligner(cx_index, wide=wide, pac=pac)
    for name, seq in cx_queries:
        ora.set_query(seq); ora.run_to(8); want = ora.blocks(with_aln=True)
        g.align_contig(seq); got = g.blocks_as_dump(with_aln=True)
        for k, v in want.items():
            assert np.array_equal(got[k], v), (name, k)
    g.close()


def test_reserved_index_memory_is_adopted_or_released(cx_index, cx_queries, ora):
    """gsa_reserve_index: device memory for the dense SA and the k-mer table set aside before the index files are read (sizes from the text length alone:
    bwt_t::seq_len, structure.h:28-38); the following gsa_create adopts what fits and frees the rest -- results are what they are without a reservation, and a
    reservation nobody adopts is freed by gsa_release_reserved."""
    import ctypes as C
    lib = capi.load_library()
    lib.gsa_reserve_index.argtypes = [C.c_int, C.c_uint64, C.c_uint32]
    lib.gsa_release_reserved.argtypes = [C.c_int]; lib.gsa_release_reserved.restype = None
    assert lib.gsa_reserve_index(0, int(cx_index.seq_len), 0) == 0
    g = capi.Aligner(cx_index)
    name, seq = cx_queries[0]
    ora.set_query(seq); ora.run_to(8); want = ora.blocks(with_aln=True)
    g.align_contig(seq); got = g.blocks_as_dump(with_aln=True)
    for k, v in want.items():
        assert np.array_equal(got[k], v), k
    g.close()
    assert lib.gsa_reserve_index(0, int(cx_index.seq_len), 1) == 0      # (wide layout sizes; never adopted)
    lib.gsa_release_reserved(0)
    assert lib.gsa_reserve_index(99, 1000, 0) != 0                        # bad device


def test_degenerate_queries(gpu, ora, golden_dir):
    """Edge cases against the committed index: empty-ish, ambiguous, unrelated, exact-copy and chunk-edge queries."""
    refs = synth.read_fasta(os.path.join(golden_dir, "cx.ref.fa"))
    ref0 = refs[0][1]
    rng = np.random.default_rng(77)
    L = min(ref0.size, 60000)
    cases = {
        "one_base": np.frombuffer(b"A", np.uint8),
        "shorter_than_a_seed": ref0[100:110].copy(),
        "all_N": np.full(25000, ord("N"), np.uint8),
        "unrelated": synth.random_genome(30000, rng),
        "exact_copy_no_gaps": ref0[:L].copy(),                             # no gap record, no DP job, empty string pools
        "exact_copy_reverse_strand": synth.revcomp(ref0[:L]),
        "chunk_edge_10000": ref0[500:10500].copy(),
        "chunk_edge_10001": ref0[500:10501].copy(),
        "chunk_edge_19999": ref0[500:20499].copy(),
        "lower_case_and_N_runs": None,
        "one_long_indel": np.concatenate([ref0[:20000], ref0[20900:L]]),   # 900-bp deletion: a striped DP job or a block cut
        "snv_every_200": None,
    }
    m = ref0[:L].copy(); m[5000:5050] = ord("N"); m[7000:9000] = np.frombuffer(bytes(m[7000:9000]).lower(), np.uint8); cases["lower_case_and_N_runs"] = m
    m = ref0[:L].copy(); idx = np.arange(100, L, 200); m[idx] = np.where(m[idx] == ord("A"), ord("C"), ord("A")); cases["snv_every_200"] = m
    for name, seq in cases.items():
        seq = np.ascontiguousarray(seq, np.uint8)
        ora.set_query(seq); gpu.set_query(seq)
        assert_stage_equal(gpu.dump_stages(8), ora.dump_stages(8), prefix="")
        ora.set_query(seq); ora.run_to(8); want = ora.blocks(with_aln=True)
        gpu.align_contig(seq); got = gpu.blocks_as_dump(with_aln=True)
        for k, v in want.items():
            assert np.array_equal(got[k], v), (name, k)


def _check_result_invariants(idx, qry, r):
    """Size-independent properties of a finished result (sizes the oracle cannot reach): records tile their block, seeds are
    exact matches, gapped strings spell exactly the two fragments, lengths and scores are what the strings say."""
    NT4 = np.full(256, 4, np.uint8)
    for k, ch in enumerate("ACGT"):
        NT4[ord(ch)] = k; NT4[ord(ch.lower())] = k
    text = idx.ref                                      # 2G ASCII bases: forward strand, then its reverse complement (indexio.unpack_pac)
    assert text.size == 2 * idx.G
    B, F, a1, a2 = r["blocks"], r["frags"], r["aln1"], r["aln2"]
    assert B.size > 0
    # records of a block are consecutive and tile it in both sequences
    for b in B:
        f = F[b["frag_off"]:b["frag_off"] + b["n_frag"]]
        assert np.array_equal(f["qpos"][1:], (f["qpos"] + f["qlen"])[:-1]) and np.array_equal(f["rpos"][1:], (f["rpos"] + f["rlen"])[:-1])
        seed = f["bseed"] != 0
        assert b["aln_len"] == int(f["qlen"][seed].sum()) + int(f["aln_len"][~seed].sum())
    used = np.concatenate([np.arange(o, o + n) for o, n in zip(B["frag_off"], B["n_frag"])])
    Fu = F[used]
    seed = Fu["bseed"] != 0
    # seeds: equal lengths, identical bases (case-insensitive)
    S = Fu[seed]
    assert np.array_equal(S["qlen"], S["rlen"])
    def gather(arr, pos, ln):
        ln = ln.astype(np.int64); tot = int(ln.sum())
        if tot == 0:
            return np.zeros(0, arr.dtype)
        starts = np.repeat(pos.astype(np.int64) - np.concatenate([[0], np.cumsum(ln)[:-1]]), ln)
        return arr[starts + np.arange(tot, dtype=np.int64)]
    assert np.array_equal(NT4[gather(qry, S["qpos"], S["qlen"])], NT4[gather(text, S["rpos"], S["rlen"])])
    # gaps: the two strings, gap characters removed, spell the reference and the query fragment
    Gp = Fu[~seed]
    s1 = gather(a1, Gp["aln_off"], Gp["aln_len"]); s2 = gather(a2, Gp["aln_off"], Gp["aln_len"])
    assert s1.size == s2.size and not np.any((s1 == ord("-")) & (s2 == ord("-")))
    assert np.array_equal(s1[s1 != ord("-")], gather(text, Gp["rpos"], Gp["rlen"]))
    assert np.array_equal(s2[s2 != ord("-")], gather(qry, Gp["qpos"], Gp["qlen"]))
    # block score = seed bases + identical columns of the gaps (CountIdenticalPairs, ProcessCandidateAlignment.cpp:38-47)
    ident = (NT4[s1] == NT4[s2]).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(ident)])
    ge = np.cumsum(Gp["aln_len"].astype(np.int64)); gb = ge - Gp["aln_len"]
    gap_score = csum[ge] - csum[gb]
    per_frag = np.zeros(Fu.size, np.int64); per_frag[seed] = S["qlen"]; per_frag[~seed] = gap_score
    fe = np.cumsum(B["n_frag"].astype(np.int64)); fb = fe - B["n_frag"]
    cs = np.concatenate([[0], np.cumsum(per_frag)])
    assert np.array_equal(cs[fe] - cs[fb], B["score"].astype(np.int64))


@pytest.mark.parametrize("total,ncontig,div,seed,repeats", [(24000000, 2, 0.015, 31, False), (250000000, 1, 0.01, 32, True), (250000000, 1, 0.01, 33, "adversarial"), (250000000, 1, 0.01, 34, "human_like")])
def test_full_size_result_invariants(tmp_path, total, ncontig, div, seed, repeats):
    """BASELINE-sized pairs (configs[3]: one 250 Mb chromosome at 1 %, with the repeat-stress injection, and once with the
    adversarial one: copy-number spectrum up to 10^5, microsatellites, Mb-long N runs, soft-masked blocks).  The 250 Mb repeat-stress
    contig is ALSO compared with the oracle at every stage (round 5: the real reference at one thread needs about a minute for it); the
    24 Mb two-strand pair and the adversarial contig (whose repeats cost the reference's walk O(L^2) Occ steps per copy: hours) are checked
    against themselves and the inputs."""
    from gsalign_amd import hostlib
    if repeats == "adversarial":
        refs, qrys = synth.make_adversarial_pair(total, ncontig, div, seed=seed)
    elif repeats == "human_like":
        refs, qrys = synth.make_human_like_pair(total, ncontig, div, seed=seed)
    else:
        refs, qrys = synth.make_pair_fast(total, ncontig, div, seed=seed, repeats=repeats)
    if ncontig > 1:
        qrys[-1] = (qrys[-1][0], synth.revcomp(qrys[-1][1]))
    rf, px = str(tmp_path / "r.fa"), str(tmp_path / "r")
    synth.write_fasta(rf, refs); hostlib.build_index(rf, px)
    idx = indexio.load_index(px)
    g = capi.Aligner(idx)
    if total == 250000000 and repeats == "human_like":
        # round 6: the primate-like repeat spectrum (the bench's `human_like` workload) at the size the bench runs it, against the REAL reference at every
        # stage.  The reference's walk pays for repeats (bwt_search.cpp:177-182): it gets ten minutes; if it has not finished, the contig's first 60 Mb are compared instead
        from oracle import oracle_py as op
        if op.have_ref():
            import subprocess
            qfa, npz = str(tmp_path / "q.fa"), str(tmp_path / "ref.npz")
            synth.write_fasta(qfa, qrys)
            proc = op.ref_dump_subprocess(px, qfa, npz, {}, upto=8, wait=False)
            q_cmp = qrys[0][1]
            try:
                assert proc.wait(timeout=600) == 0
            except subprocess.TimeoutExpired:
                proc.kill(); proc.wait()
                q_cmp = np.ascontiguousarray(qrys[0][1][:60000000])
                synth.write_fasta(qfa, [("q60", q_cmp)]); op.ref_dump_subprocess(px, qfa, npz, {}, upto=8)
            z = np.load(npz); want = {k[3:]: z[k] for k in z.files if k.startswith("c0_")}
            g.set_query(q_cmp)
            assert_stage_equal(g.dump_stages(8), want)
            assert want["s8_b_score"].size > 0 and want["s1_qpos"].size > 200000
            print(f"human-like {q_cmp.size} bp vs the real reference: {want['s1_qpos'].size} seeds, {want['s8_b_score'].size} blocks, {want['s8_f_qpos'].size} records, {want['s8_aln1'].size} string bytes per side -- identical at every stage")
    if total == 250000000 and repeats is True:
        # BASELINE configs[3] against the oracle (round 5): the whole 250 Mb contig through the real reference at one thread (libgsref,
        # ~60 s; the CPU restatement where oracle/_ref is absent), every stage dump S1..S8 incl. both gapped-string pools, bit for bit
        from oracle import oracle_py as op
        if op.have_ref():
            qfa, npz = str(tmp_path / "q.fa"), str(tmp_path / "ref.npz")
            synth.write_fasta(qfa, qrys); op.ref_dump_subprocess(px, qfa, npz, {}, upto=8)
            z = np.load(npz); want = {k[3:]: z[k] for k in z.files if k.startswith("c0_")}
        else:
            op.build(ref=False); o = op.Oracle(idx); o.set_query(qrys[0][1]); want = o.dump_stages(8); o.close()
        g.set_query(qrys[0][1])
        assert_stage_equal(g.dump_stages(8), want)
        assert want["s8_b_score"].size > 0 and want["s1_qpos"].size > 1000000
        print(f"250 Mb contig vs the {'real reference' if op.have_ref() else 'oracle restatement'}: {want['s1_qpos'].size} seeds, {want['s8_b_score'].size} blocks, {want['s8_f_qpos'].size} records, {want['s8_aln1'].size} string bytes per side -- identical at every stage")
    for name, seq in qrys:
        g.align_contig(seq)
        r = g.blocks()
        _check_result_invariants(idx, seq, r)
        cov = int(r["blocks"]["aln_len"].sum())
        assert cov > (0.8 if repeats in ("adversarial", "human_like") else 0.9) * seq.size, (name, cov)
        if repeats == "adversarial":
            print(f"adversarial 250 Mb: {int(g.seed_stats()[1])} of {(seq.size + 9999) // 10000} chunks took the dense search, {r['blocks'].size} blocks, coverage {cov / seq.size:.3f}")
    g.close()


def test_config5_full_human_all_contigs():
    """BASELINE configs[4] on one GPU (the whole job of the 8-GPU configuration; the index is replicated per GPU there): a 24-contig
    reference with GRCh38 chromosome lengths, 3.08 Gbp / 6.2 G BWT rows -- the >= 2^32-row device layout and the 64-bit suffix sorter
    on their real input (the index builds in ~80 s on the host's cores since round 3) -- vs a 1 %-diverged query, -alen 5000, through
    gsa_align_many on two contexts, then EVERY contig through the result invariants, and -- round 5 -- ORACLE PARITY at this scale: two whole
    contigs (46 Mb reverse strand, 58 Mb forward), a 20 Mb reverse-strand piece of chr1 (reference positions above 2^32) and four short
    pieces against the real reference (libgsref loads the same index files: bwt_index.cpp:147-264), every stage dump S1..S8 with both string
    pools, bit for bit; round 6 -- ALL 24 contigs against the real reference at stages 1 and 8 (seeds, groups, blocks, records, both string pools).  tests/human_scale_check.py; needs a host with >= 256 GB of memory (skipped elsewhere)."""
    import subprocess
    import sys
    from conftest import ROOT
    mem_gb = 0.0
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemTotal"):
            mem_gb = int(ln.split()[1]) / 1e6
    if mem_gb < 256:
        pytest.skip(f"host has {mem_gb:.0f} GB of memory")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "human_scale_check.py")], capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0 and "HUMAN SCALE PROBE OK" in r.stdout and "all 24 contigs checked" in r.stdout and "ORACLE PARITY OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    from oracle import oracle_py as op
    if op.have_ref():      # round 6: every contig of the genome against the real reference (stages 1 and 8), reference processes side by side on the host's cores
        assert "WHOLE GENOME == real reference: all 24 contigs" in r.stdout, r.stdout[-3000:]
    print(r.stdout)


def test_align_many_over_two_devices(oracle_built, tmp_path):
    """The day a box has two GPUs: gsa_align_many with one context on each (gsa_create per device: the index replicated) --
    contigs dealt over both, and ONE long contig seeded by chunk range on both with the hits moved peer to peer (hipMemcpyPeerAsync,
    gsa_import_hits) -- against the oracle.  Skipped on one-GPU boxes (every box this round): nothing here has run on two devices yet."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (hipGetDeviceCount() >= 2)")
    refs, qrys = synth.make_pair_fast(2400000, 4, 0.02, seed=91)
    idx = _build(tmp_path, refs)
    o = oracle_built.Oracle(idx)
    want = []
    for name, seq in qrys:
        o.set_query(seq); o.run_to(8); want.append(o.blocks(with_aln=True))
    o.close()
    ctxs = [capi.Aligner(idx, device=0), capi.Aligner(idx, device=1)]
    ctxs[0].set_option("split_min", 200000)
    got = {}

    def on_result(ci, res):
        got[ci] = capi.result_as_dump(ctxs[0]._result(res), with_aln=True)
        return 0
    capi.align_many(ctxs, [q for _, q in qrys], on_result)                       # four contigs on two devices
    assert sorted(got) == list(range(len(qrys)))
    for ci, d in got.items():
        for key, v in want[ci].items():
            assert np.array_equal(d[key], v), (ci, key)
    got.clear()
    capi.align_many(ctxs, [qrys[0][1]], on_result)                               # one contig: seeded on both, hits peer to peer
    for key, v in want[0].items():
        assert np.array_equal(got[0][key], v), ("split", key)
    for g in ctxs:
        g.close()
