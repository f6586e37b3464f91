"""The CPU restatement (oracle/gsa_oracle.cpp) against the committed golden
vectors, all of which were produced by the REAL reference (see
tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_stage_equal


@pytest.fixture(scope="module")
def ora(oracle_built, cx_index):
    o = oracle_built.Oracle(cx_index)
    yield o
    o.close()


def test_index_loader_shapes(cx_index):
    G = cx_index.G
    assert cx_index.seq_len == 2 * G
    assert cx_index.bwt.size == (2 * G + 15) // 16 + 8 * ((2 * G + 127) // 128 + 1)
    assert cx_index.sa.size == (2 * G + 32) // 32
    assert cx_index.ref.size == 2 * G and int(cx_index.chr_len.sum()) == G


def test_stages_default_all_contigs(ora, cx_queries):
    want = np.load(os.path.join(GOLDEN, "cx_stages.npz"))
    ora.set_params()
    for ci, (name, seq) in enumerate(cx_queries):
        ora.set_query(seq)
        got = ora.dump_stages(8)
        assert_stage_equal(got, want, prefix=f"c{ci}_")
        assert ora.counters()[7] == 0, "fixture hit the reference's undefined whole-group-died case"


def test_stages_sensitive(ora, cx_queries):
    want = np.load(os.path.join(GOLDEN, "cx_sen_stages.npz"))
    ora.set_params(sen=1, clr=50)
    for ci, (name, seq) in enumerate(cx_queries):
        ora.set_query(seq)
        assert_stage_equal(ora.dump_stages(8), want, prefix=f"c{ci}_")
    ora.set_params()


def test_ksw2_known_answers(oracle_built):
    d = np.load(os.path.join(GOLDEN, "ksw2_pairs.npz"))
    n = d["s1_off"].size - 1
    assert n >= 2000
    for i in range(n):
        s1 = d["s1"][d["s1_off"][i]:d["s1_off"][i + 1]].tobytes(); s2 = d["s2"][d["s2_off"][i]:d["s2_off"][i + 1]].tobytes()
        a1 = d["a1"][d["a1_off"][i]:d["a1_off"][i + 1]].tobytes(); a2 = d["a2"][d["a2_off"][i]:d["a2_off"][i + 1]].tobytes()
        o1, o2 = oracle_built.oracle_ksw2(s1, s2)
        assert (o1, o2) == (a1, a2), f"pair {i}: {s1[:40]} / {s2[:40]}"


def test_ksw2_survey_examples(oracle_built):
    # verified against the reference while surveying
    k = oracle_built.oracle_ksw2
    assert k(b"ACGTACGTTTGACCA", b"ACGTACGTGACCA") == (b"ACGTACGTTTGACCA", b"ACGTACG--TGACCA")
    assert k(b"AAAAACCCCC", b"AAAAAGCCCCC") == (b"AAAAA-CCCCC", b"AAAAAGCCCCC")
    assert k(b"A", b"ACGT") == (b"A---", b"ACGT")
    assert k(b"ACGT", b"TTTT") == (b"ACGT", b"TTTT")


def test_gap_similarity_known_answers(ora, cx_queries):
    rows = np.load(os.path.join(GOLDEN, "gapsim.npz"))["rows"]
    cur = -1
    for ci, q1, q2, r1, r2, want in rows:
        if ci != cur:
            ora.set_query(cx_queries[int(ci)][1]); cur = ci
        assert ora.gap_similarity(int(q1), int(q2), int(r1), int(r2)) == want
