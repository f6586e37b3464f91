"""numpy reader for the BWA-style index files GSAlign uses (.bwt .sa .ann .pac).

Python-side mirror of the reference loader (reference src/bwt_index.cpp:25-264);
layouts are in SURVEY.md Appendix C.  Used by tests and bench.py to hand the
index to the C-ABI (`gsa_index_view`) as plain host arrays.  The C++ host
program has its own loader (gsalign_amd/csrc/host/index_io.cpp).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class BwaIndex:
    hdr: np.ndarray        # uint64[5]: primary, L2[1..4]
    bwt: np.ndarray        # uint32[]: interleaved Occ + BWT words
    sa: np.ndarray         # uint64[n_sa], sa[0] = 2**64-1
    G: int                 # forward length (bns->l_pac)
    chr_names: list
    chr_len: np.ndarray    # int32[n_chr]
    ref: np.ndarray        # uint8[2G] ASCII, forward + reverse complement

    @property
    def seq_len(self) -> int:
        return int(self.hdr[4])


def unpack_pac(pac: np.ndarray, G: int) -> np.ndarray:
    """2-bit forward pac -> 2G ASCII bytes (bwt_index.cpp:193-209)."""
    shifts = np.array([6, 4, 2, 0], dtype=np.uint8)
    codes = ((pac[:, None] >> shifts[None, :]) & 3).reshape(-1)[:G]
    fwd = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    rev = np.frombuffer(b"TGCA", dtype=np.uint8)[codes][::-1]
    return np.concatenate([fwd, rev])


def load_index(prefix: str) -> BwaIndex:
    raw = np.fromfile(prefix + ".bwt", dtype=np.uint8)
    hdr = raw[:40].view(np.uint64).copy()
    bwt = raw[40:].view(np.uint32).copy()
    seq_len = int(hdr[4])
    sraw = np.fromfile(prefix + ".sa", dtype=np.uint64)
    assert int(sraw[0]) == int(hdr[0]) and int(sraw[6]) == seq_len and int(sraw[5]) == 32
    n_sa = (seq_len + 32) // 32
    sa = np.empty(n_sa, dtype=np.uint64)
    sa[0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    sa[1:] = sraw[7:7 + n_sa - 1]
    with open(prefix + ".ann", "r") as fh:
        toks = fh.readline().split()
        G, n_seqs = int(toks[0]), int(toks[1])
        names, lens = [], []
        for _ in range(n_seqs):
            names.append(fh.readline().split()[1])
            lens.append(int(fh.readline().split()[1]))
    pac = np.fromfile(prefix + ".pac", dtype=np.uint8)[: G // 4 + 1]
    ref = unpack_pac(pac, G)
    assert seq_len == 2 * G
    return BwaIndex(hdr, bwt, sa, G, names, np.asarray(lens, dtype=np.int32), ref)
