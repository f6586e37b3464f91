"""CPU-side checks of the C-ABI library: it loads, exports every symbol
include/gsa_hip.h declares, and refuses to run without a GPU (no CPU path)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from gsalign_amd import capi


@pytest.fixture(scope="module")
def lib():
    capi.build_library()
    return capi.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "gsa_hip.h")).read()
    declared = set(re.findall(r"\b(gsa_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"gsa_ctx"}
    declared -= set(re.findall(r"static inline [a-z0-9_ ]+\b(gsa_[a-z0-9_]+)\s*\(", hdr))      # (helpers defined in the header itself)
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in sorted(declared):
        assert getattr(lib, name) is not None


def test_struct_layouts_match_header():
    assert C.sizeof(capi.Seed) == 16 and C.sizeof(capi.Frag) == 40 and C.sizeof(capi.Block) == 40
    assert capi.FRAG_DT.itemsize == 40 and capi.BLOCK_DT.itemsize == 40 and capi.SEED_DT.itemsize == 16
    assert C.sizeof(capi.Rec) == 16 and capi.REC_DT.itemsize == 16


def test_compact_records_expand_like_the_header(tmp_path):
    """gsa_rec (16 bytes: what crosses PCIe) -> gsa_frag: the header's inline gsa_expand_frags, compiled here as plain C, against
    capi.expand_recs on random blocks of seed [gap] seed ... records; pack_recs is the inverse."""
    import subprocess
    import numpy as np
    src = tmp_path / "x.c"
    src.write_text('#include "gsa_hip.h"\nvoid expand(const gsa_rec *r, int64_t n, gsa_frag *o) { gsa_expand_frags(r, n, o); }\nint is_seed(const gsa_rec *r) { return gsa_rec_is_seed(r); }\n')
    so = tmp_path / "x.so"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(so)], check=True)
    x = C.CDLL(str(so))
    rng = np.random.default_rng(5)
    F = []
    for _ in range(40):                                   # blocks
        q, r = int(rng.integers(0, 1 << 20)), int(rng.integers(0, 1 << 40))
        for k in range(int(rng.integers(1, 30))):
            ln = int(rng.integers(10, 500))
            F.append((1, q, ln, ln, r, 0, 0, 0)); q += ln; r += ln
            if rng.random() < 0.7:
                gq, gr = int(rng.integers(0, 50)), int(rng.integers(0, 50))
                F.append((0, q, gq, gr, r, int(rng.integers(0, 1 << 31)), int(rng.integers(0, 100)), 0)); q += gq; r += gr
        if F[-1][0] == 0:
            F.pop()
    F = np.array(F, dtype=capi.FRAG_DT)
    R = capi.pack_recs(F)
    assert R.itemsize == 16 and np.array_equal(capi.expand_recs(R), F)
    out = np.zeros(F.size, capi.FRAG_DT)
    x.expand(C.c_void_p(R.ctypes.data), C.c_int64(R.size), C.c_void_p(out.ctypes.data))
    assert np.array_equal(out, F)
    assert [x.is_seed(C.c_void_p(R[i:i + 1].ctypes.data)) for i in range(50)] == [int(b) for b in F["bseed"][:50]]


def test_no_cpu_fallback(lib, cx_index):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.GsaError, match="no HIP device"):
        capi.Aligner(cx_index)
