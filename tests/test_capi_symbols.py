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
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in sorted(declared):
        assert getattr(lib, name) is not None


def test_struct_layouts_match_header():
    assert C.sizeof(capi.Seed) == 16 and C.sizeof(capi.Frag) == 40 and C.sizeof(capi.Block) == 40
    assert capi.FRAG_DT.itemsize == 40 and capi.BLOCK_DT.itemsize == 40 and capi.SEED_DT.itemsize == 16


def test_no_cpu_fallback(lib, cx_index):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.GsaError, match="no HIP device"):
        capi.Aligner(cx_index)
