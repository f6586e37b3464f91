"""Shared fixtures.  `-m gpu` tests need a real MI355X; everything else runs on CPU."""
import gzip
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def gunzip_to(src, dst):
    with gzip.open(src, "rb") as a, open(dst, "wb") as b:
        shutil.copyfileobj(a, b)


@pytest.fixture(scope="session")
def golden_dir(tmp_path_factory):
    """tests/golden/*.gz unpacked into a scratch directory (index files, FASTA, MAF, VCF)."""
    d = tmp_path_factory.mktemp("golden")
    for fn in os.listdir(GOLDEN):
        if fn.endswith(".gz"):
            gunzip_to(os.path.join(GOLDEN, fn), os.path.join(d, fn[:-3]))
    return str(d)


@pytest.fixture(scope="session")
def cx_index(golden_dir):
    from gsalign_amd import indexio
    return indexio.load_index(os.path.join(golden_dir, "cx"))


@pytest.fixture(scope="session")
def cx_queries(golden_dir):
    from gsalign_amd import synth
    return synth.read_fasta(os.path.join(golden_dir, "cx.qry.fa"))


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import oracle_py as op
    # `make` only where something is stale: the GPU box gets the built libraries with the snapshot and need not have a compiler (build() in
    # __graft_entry__ made them; /root/reference does not exist there, so oracle/_ref is never rebuilt on it)
    src = [os.path.join(ROOT, "oracle", f) for f in ("gsa_oracle.cpp", "gsa_oracle.h", "Makefile")]
    fresh = os.path.exists(op.ORACLE_SO) and all(os.path.getmtime(op.ORACLE_SO) >= os.path.getmtime(f) for f in src if os.path.exists(f))
    have_reference = os.path.isdir("/root/reference")
    if have_reference or not fresh:      # (here: make is incremental and also keeps oracle/_ref current; on the GPU box: only when the snapshot's library is stale)
        op.build(ref=have_reference)
    return op


def assert_stage_equal(got: dict, want, prefix: str = "", stages=range(1, 9)):
    """Compare two stage dumps key by key (exact)."""
    for k, v in got.items():
        st = int(k[1:k.index("_")])
        if st not in stages:
            continue
        w = want[prefix + k]
        assert w.shape == v.shape, f"{prefix}{k}: shape {v.shape} != golden {w.shape}"
        assert np.array_equal(w, v), f"{prefix}{k}: values differ (first at {np.flatnonzero(w != v)[:5]})"
