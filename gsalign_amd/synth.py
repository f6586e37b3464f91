"""Synthetic genomes for tests and bench.py.

Reference: i.i.d. uniform ACGT contigs.  Query: the reference with per-base
events at total rate ``d`` -- 80 % substitutions (uniform over the three other
bases), 10 % insertions of length U[1,10], 10 % deletions of length U[1,10].
Everything is vectorised numpy so that a 5 Mb pair takes well under a second.

This is workload tooling, not part of the aligner.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


def random_genome(n: int, rng: np.random.Generator) -> np.ndarray:
    """n uniform random bases as ASCII uint8."""
    return _ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


def mutate(ref: np.ndarray, d: float, rng: np.random.Generator) -> np.ndarray:
    """Apply the SNV/indel event mix at total per-base rate ``d``."""
    n = ref.size
    if d <= 0:
        return ref.copy()
    ev = rng.random(n)
    sub = ev < 0.8 * d
    ins = (ev >= 0.8 * d) & (ev < 0.9 * d)
    dele = (ev >= 0.9 * d) & (ev < d)
    out = ref.copy()
    # substitutions: rotate by 1..3 within ACGT
    code = np.zeros(256, dtype=np.uint8)
    code[_ACGT] = np.arange(4, dtype=np.uint8)
    k = int(sub.sum())
    out[sub] = _ACGT[(code[ref[sub]] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3]
    # deletions: drop L bases starting at the event position
    keep = np.ones(n, dtype=bool)
    dpos = np.flatnonzero(dele)
    dlen = rng.integers(1, 11, size=dpos.size)
    for off in range(10):
        sel = dpos[dlen > off] + off
        keep[sel[sel < n]] = False
    # insertions: L random bases in front of the event position
    rep = keep.astype(np.int64)
    ipos = np.flatnonzero(ins)
    ilen = rng.integers(1, 11, size=ipos.size)
    rep[ipos] += ilen
    idx = np.repeat(np.arange(n), rep)
    res = out[idx]
    # inside each repeated run the last copy is the original base (if kept);
    # the copies before it are inserted bases -> randomise them
    first = np.ones(idx.size, dtype=bool)
    first[1:] = idx[1:] != idx[:-1]
    run_start = np.flatnonzero(first)
    run_len = np.diff(np.append(run_start, idx.size))
    run_id = np.cumsum(first) - 1
    pos_in_run = np.arange(idx.size) - run_start[run_id]
    n_ins = run_len[run_id] - keep[idx].astype(np.int64)
    is_ins = pos_in_run < n_ins
    res[is_ins] = random_genome(int(is_ins.sum()), rng)
    return res


# ---------------------------------------------------------------------------
# The same generator family in C++ (csrc/host/synth.cpp, counter-based RNG): a 250 Mb pair in seconds.
# ---------------------------------------------------------------------------
def _hostlib():
    import ctypes as C
    from . import hostlib
    lib = hostlib.load()
    lib.gsah_c_synth_genome.argtypes = [C.c_int64, C.c_uint64, C.c_void_p]
    lib.gsah_c_synth_repeats.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int]
    lib.gsah_c_synth_repeats.restype = C.c_int64
    lib.gsah_c_synth_mutate.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_uint64, C.c_void_p, C.c_int64]
    lib.gsah_c_synth_mutate.restype = C.c_int64
    lib.gsah_c_synth_adversarial.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int, C.c_int64, C.c_int64]
    lib.gsah_c_synth_adversarial.restype = C.c_int64
    lib.gsah_c_synth_human_like.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int64]
    lib.gsah_c_synth_human_like.restype = C.c_int64
    return lib


def fast_genome(n: int, seed: int = 11) -> np.ndarray:
    out = np.empty(n, np.uint8)
    _hostlib().gsah_c_synth_genome(n, seed, out.ctypes.data)
    return out


def inject_repeats(seq: np.ndarray, seed: int = 11, frac: float = 0.10, fam_len: int = 300, copy_div: float = 0.10,
                   tandem_unit: int = 40, tandem_copies: int = 150) -> int:
    """Repeat-stress variant of SURVEY.md section 8(d), in place: a 300-bp family (copies 10 % divergent) covering `frac`
    of the genome plus one tandem array with more than MaxSeedFreq copies.  Returns the number of family copies."""
    assert seq.flags.c_contiguous and seq.dtype == np.uint8
    return int(_hostlib().gsah_c_synth_repeats(seq.ctypes.data, seq.size, seed, frac, fam_len, copy_div, tandem_unit, tandem_copies))


def inject_adversarial(seq: np.ndarray, seed: int = 11, frac: float = 0.25, n_fam: int = 8, max_copies: int = 100_000, n_run: int = 1_000_000) -> int:
    """In place: repeat families with a copy-number spectrum (up to max_copies copies, 1-15 % divergent), microsatellites,
    two runs of n_run N's (capped at 1/8 of the sequence) and soft-masked blocks -- csrc/host/synth.cpp.  Returns the copy count."""
    assert seq.flags.c_contiguous and seq.dtype == np.uint8
    return int(_hostlib().gsah_c_synth_adversarial(seq.ctypes.data, seq.size, seed, frac, n_fam, max_copies, min(n_run, seq.size // 9)))


def inject_human_like(seq: np.ndarray, seed: int = 11, scale: float = 1.0, n_run: int = 1_000_000) -> int:
    """In place: the interspersed-repeat spectrum of a primate genome over ~45 % of the sequence (x `scale`) -- an Alu-like family in three age
    classes, 5'-truncated L1-like copies, LTR-like families with solo LTRs, ancient repeats, segmental duplications, microsatellites, soft-masked
    blocks and two N runs -- csrc/host/synth.cpp.  Returns the copy count."""
    assert seq.flags.c_contiguous and seq.dtype == np.uint8
    return int(_hostlib().gsah_c_synth_human_like(seq.ctypes.data, seq.size, seed, scale, min(n_run, seq.size // 9)))


def make_human_like_pair(total_len: int, n_contigs: int, d: float, seed: int = 11, **kw):
    """(ref_contigs, qry_contigs) like make_pair_fast, every reference contig with the human-like repeat spectrum."""
    base = total_len // n_contigs
    refs, qrys = [], []
    for i in range(n_contigs):
        ln = base if i + 1 < n_contigs else total_len - base * (n_contigs - 1)
        r = fast_genome(int(ln), seed * 1000 + i)
        inject_human_like(r, seed * 1000 + i, **kw)
        refs.append((f"chr{i + 1}", r))
        qrys.append((f"qry{i + 1}", fast_mutate(r, d, seed * 1000 + 500 + i)))
    return refs, qrys


def make_adversarial_pair(total_len: int, n_contigs: int, d: float, seed: int = 11, **kw):
    """(ref_contigs, qry_contigs) like make_pair_fast, every reference contig with the adversarial injection."""
    base = total_len // n_contigs
    refs, qrys = [], []
    for i in range(n_contigs):
        ln = base if i + 1 < n_contigs else total_len - base * (n_contigs - 1)
        r = fast_genome(int(ln), seed * 1000 + i)
        inject_adversarial(r, seed * 1000 + i, **kw)
        refs.append((f"chr{i + 1}", r))
        qrys.append((f"qry{i + 1}", fast_mutate(r, d, seed * 1000 + 500 + i)))
    return refs, qrys


def fast_mutate(ref: np.ndarray, d: float, seed: int) -> np.ndarray:
    ref = np.ascontiguousarray(ref, np.uint8)
    cap = ref.size + ref.size // 32 + 4096
    out = np.empty(cap, np.uint8)
    n = int(_hostlib().gsah_c_synth_mutate(ref.ctypes.data, ref.size, d, seed, out.ctypes.data, cap))
    assert n >= 0
    return out[:n]


def make_pair_fast(total_len: int, n_contigs: int, d: float, seed: int = 11, repeats: bool = False, lengths=None):
    """(ref_contigs, qry_contigs) from the C++ generator; `lengths` overrides the equal split."""
    if lengths is None:
        base = total_len // n_contigs
        lengths = [base] * (n_contigs - 1) + [total_len - base * (n_contigs - 1)]
    refs, qrys = [], []
    for i, ln in enumerate(lengths):
        r = fast_genome(int(ln), seed * 1000 + i)
        if repeats:
            inject_repeats(r, seed * 1000 + i)
        refs.append((f"chr{i + 1}", r))
        qrys.append((f"qry{i + 1}", fast_mutate(r, d, seed * 1000 + 500 + i)))
    return refs, qrys


def write_fasta(path: str, contigs: list[tuple[str, np.ndarray]], width: int = 70) -> None:
    with open(path, "wb") as fh:
        for name, seq in contigs:
            fh.write(b">" + name.encode() + b"\n")
            n = seq.size
            full = n // width
            if full:
                body = np.empty((full, width + 1), dtype=np.uint8)
                body[:, :width] = seq[: full * width].reshape(full, width)
                body[:, width] = 10
                fh.write(body.tobytes())
            if n % width:
                fh.write(seq[full * width:].tobytes() + b"\n")


def read_fasta(path: str) -> list[tuple[str, np.ndarray]]:
    """Minimal FASTA reader for tests (name = text up to first whitespace)."""
    out: list[tuple[str, list[bytes]]] = []
    with open(path, "rb") as fh:
        for line in fh:
            line = line.rstrip(b"\r\n")
            if not line:
                continue
            if line[:1] == b">":
                out.append((line[1:].split()[0].decode() if line[1:].split() else "", []))
            else:
                out[-1][1].append(line)
    return [(n, np.frombuffer(b"".join(p), dtype=np.uint8).copy()) for n, p in out]


def make_pair(total_len: int, n_contigs: int, d: float, seed: int = 11):
    """(ref_contigs, qry_contigs): n_contigs contigs totalling total_len, query = mutated copy."""
    rng = np.random.default_rng(seed)
    base = total_len // n_contigs
    refs, qrys = [], []
    for i in range(n_contigs):
        ln = base if i < n_contigs - 1 else total_len - base * (n_contigs - 1)
        r = random_genome(ln, rng)
        refs.append((f"chr{i + 1}", r))
        qrys.append((f"qry{i + 1}", mutate(r, d, rng)))
    return refs, qrys


# ---------------------------------------------------------------------------
# "complex" fixture: exercises every branch of the chaining heuristics
# (SURVEY.md section 8(c) F1/F4/F5 and Appendix A.8 "cx", "hd", "gap").
# ---------------------------------------------------------------------------
def _place(seq: np.ndarray, pos: int, piece: np.ndarray) -> None:
    seq[pos:pos + piece.size] = piece


def make_complex(seed: int = 2024, scale: int = 1):
    """Returns (ref_contigs, qry_contigs).  ~270 kb reference, 8 query contigs."""
    rng = np.random.default_rng(seed)
    c1 = random_genome(120000 * scale, rng)
    c2 = random_genome(90000 * scale, rng)
    c3 = random_genome(60000 * scale, rng)
    # 300-bp repeat family, 40 copies, 10 % copy divergence
    fam = random_genome(300, rng)
    for c in (c1, c2, c3):
        for p in rng.integers(1000, c.size - 1000, size=14):
            _place(c, int(p), mutate(fam, 0.10, rng)[:300])
    # 40-bp unit x 150 tandem copies (> MaxSeedFreq hits for every k-mer inside)
    unit = random_genome(40, rng)
    _place(c1, 70000, np.tile(unit, 150))
    # 5 kb segmental duplication chr1 -> chr2 (1 % divergent)
    _place(c2, 40000, mutate(c1[20000:25000], 0.01, rng)[:5000])
    # reference N-run (randomised by the index builder)
    ref1 = c1.copy(); ref1[50000:50200] = ord("N")
    refs = [("chrA", ref1), ("chrB", c2), ("chrC", c3)]

    qs = []
    # q1: 2 % divergence + SVs
    a = c1.copy()
    parts = [a[:15000], a[17000:30000],                      # 2 kb deletion
             random_genome(3000, rng), a[30000:40000],      # 3 kb novel insertion
             revcomp(a[40000:60000]),                       # 20 kb inversion
             a[60000:80000], random_genome(60, rng), a[80000:95000],   # +60 diagonal jump
             a[95030:120000 * scale]]                       # -30 diagonal jump
    qs.append(("q1_sv", mutate(np.concatenate(parts), 0.02, rng)))
    # q2: lower-case, with N / n / R sprinkled
    b = mutate(c2, 0.01, rng)
    b = np.frombuffer(b.tobytes().lower(), dtype=np.uint8).copy()
    for ch, k in ((ord("N"), 40), (ord("n"), 40), (ord("R"), 20)):
        b[rng.integers(0, b.size, size=k)] = ch
    b[30000:30150] = ord("N")
    qs.append(("q2_lower", b))
    # q3: contig bridging chrB tail and chrC head
    qs.append(("q3_bridge", mutate(np.concatenate([c2[-30000:], c3[:30000]]), 0.01, rng)))
    # q4: 5 % divergent with a tandem duplication
    d = np.concatenate([c3[:30000], c3[25000:30000], c3[30000:]])
    qs.append(("q4_div5", mutate(d, 0.05, rng)))
    # q5: pure reverse-strand contig
    qs.append(("q5_rev", mutate(revcomp(c1[5000:45000]), 0.01, rng)))
    # q6: divergence islands 3-12 %
    e = c2.copy(); segs = []
    for s in range(0, e.size, 3000):
        dv = 0.12 if (s // 3000) % 3 == 0 else 0.03
        segs.append(mutate(e[s:s + 3000], dv, rng))
    qs.append(("q6_islands", np.concatenate(segs)))
    # q7: gap stress -- poly-A and periodic replacements of 400..2000 bp, IUPAC inside gaps
    g = mutate(c3, 0.005, rng)
    pos = 3000
    while pos + 2500 < g.size:
        ln = int(rng.integers(400, 2000))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            g[pos:pos + ln] = ord("A")
        elif kind == 1:
            g[pos:pos + ln] = np.tile(np.frombuffer(b"ACG", dtype=np.uint8), ln // 3 + 1)[:ln]
        elif kind == 2:
            g[pos:pos + ln] = random_genome(ln, rng); g[pos + 10:pos + 20] = ord("N"); g[pos + 50] = ord("Y")
        else:
            m = mutate(g[pos:pos + ln], 0.30, rng)[:ln]; g[pos:pos + m.size] = m
        pos += ln + int(rng.integers(1500, 4000))
    qs.append(("q7_gaps", g))
    # q8: short contig below every threshold + one with no hits at all
    qs.append(("q8_tiny", c1[1000:1150].copy()))
    qs.append(("q9_nohit", random_genome(5000, rng)))
    return refs, qs
