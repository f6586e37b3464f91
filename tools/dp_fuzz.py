#!/usr/bin/env python3
"""gsa_ksw2_batch against the oracle's ksw2 restatement on random pairs of many shapes (GPU box).
Shapes are chosen around the striped kernel's edges: query lengths around multiples of 64 and 128 (one wave
takes two 64-column stripes), very short and very long reference sides, N bases, identical and unrelated pairs.
usage: dp_fuzz.py [pairs] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gsalign_amd import capi
from oracle import oracle_py


def make_pairs(npairs, seed):
    rng = np.random.default_rng(seed)
    edge = [1, 2, 3, 16, 17, 63, 64, 65, 66, 127, 128, 129, 130, 191, 192, 193, 255, 256, 257, 319, 320, 321, 383, 384, 385, 448, 511, 512, 513, 640, 700]
    s1, s2 = [], []
    for i in range(npairs):
        kind = i % 8
        n = int(edge[rng.integers(len(edge))]) if kind < 5 else int(rng.integers(1, 900))
        if kind == 0: m = int(rng.integers(1, 8))
        elif kind == 1: m = int(rng.integers(100, 140))
        elif kind == 2: m = n
        elif kind == 3: m = max(1, n + int(rng.integers(-20, 21)))
        elif kind == 4: m = int(rng.integers(1, 1500))
        else: m = max(1, int(n * rng.uniform(0.5, 1.6)))
        if m + n - 1 <= 128 and n <= 64 and i % 3: m = 129 + int(rng.integers(0, 300))      # (mostly jobs of the striped kernel)
        a = rng.integers(0, 4, m).astype(np.uint8)
        mode = int(rng.integers(0, 6))
        if mode == 0: b = rng.integers(0, 4, n).astype(np.uint8)                                # unrelated
        else:
            # b = a with substitutions and indels, cut / padded to n
            out = []; j = 0; d = (0.02, 0.08, 0.2, 0.4, 0.0)[mode - 1]
            while j < m and len(out) < n:
                r = rng.random()
                if r < d * 0.6: out.append((int(a[j]) + 1 + int(rng.integers(0, 3))) & 3); j += 1
                elif r < d * 0.8: out.extend(rng.integers(0, 4, int(rng.integers(1, 12))).tolist())
                elif r < d: j += int(rng.integers(1, 12))
                else: out.append(int(a[j])); j += 1
            out = out[:n]
            while len(out) < n: out.append(int(rng.integers(0, 4)))
            b = np.array(out, dtype=np.uint8)
        A = np.frombuffer(b"ACGT", dtype=np.uint8)[a].copy(); B = np.frombuffer(b"ACGT", dtype=np.uint8)[b].copy()
        if i % 11 == 0:
            A[rng.integers(0, m, max(1, m // 30))] = ord("N")
            B[rng.integers(0, n, max(1, n // 30))] = ord("N")
        s1.append(A.tobytes()); s2.append(B.tobytes())
    return s1, s2


def make_small_pairs(npairs, seed):
    """Pairs of the classes below the striped kernel (n <= 64 query bases, m + n - 1 <= 128): every corner of that domain -- one
    row, one column, 64 columns, 128 rows, products around the lane kernel's cell limit (512) and around its 8-cell direction
    words -- with substitutions, indels, unrelated and identical pairs and N bases."""
    rng = np.random.default_rng(seed)
    s1, s2 = [], []
    ns = [1, 2, 3, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 32, 33, 40, 48, 56, 57, 63, 64]
    for i in range(npairs):
        kind = i % 6
        n = int(ns[rng.integers(len(ns))]) if kind < 4 else int(rng.integers(1, 65))
        mmax = 129 - n
        if kind == 0: m = 1 + int(rng.integers(0, min(mmax, 4)))
        elif kind == 1: m = mmax - int(rng.integers(0, min(mmax, 3)))
        elif kind == 2: m = min(mmax, max(1, (512 + int(rng.integers(-2, 3)) * n) // n))       # m * n around 512
        elif kind == 3: m = min(mmax, max(1, n + int(rng.integers(-6, 7))))
        else: m = 1 + int(rng.integers(0, mmax))
        a = rng.integers(0, 4, m).astype(np.uint8)
        mode = int(rng.integers(0, 6))
        if mode == 0: b = rng.integers(0, 4, n).astype(np.uint8)
        else:
            out = []; j = 0; d = (0.02, 0.08, 0.2, 0.4, 0.0)[mode - 1]
            while j < m and len(out) < n:
                r = rng.random()
                if r < d * 0.6: out.append((int(a[j]) + 1 + int(rng.integers(0, 3))) & 3); j += 1
                elif r < d * 0.8: out.extend(rng.integers(0, 4, int(rng.integers(1, 6))).tolist())
                elif r < d: j += int(rng.integers(1, 6))
                else: out.append(int(a[j])); j += 1
            out = out[:n]
            while len(out) < n: out.append(int(rng.integers(0, 4)))
            b = np.array(out, dtype=np.uint8)
        A = np.frombuffer(b"ACGT", dtype=np.uint8)[a].copy(); B = np.frombuffer(b"ACGT", dtype=np.uint8)[b].copy()
        if i % 7 == 0:
            A[rng.integers(0, m, max(1, m // 10))] = ord("N")
            B[rng.integers(0, n, max(1, n // 10))] = ord("N")
        if i % 13 == 0: A = np.frombuffer(bytes(A).lower(), np.uint8).copy()
        s1.append(A.tobytes()); s2.append(B.tobytes())
    return s1, s2


def make_large_pairs(npairs, seed):
    """Few long pairs around the LDS limit of the four-wave layout (reference side 3968) up to the 5000-base gap limit."""
    rng = np.random.default_rng(seed)
    s1, s2 = [], []
    ms = [3967, 3968, 3969, 4100, 4999, 5000, 3000, 2500]
    for i in range(npairs):
        m = ms[i % len(ms)]; n = int((65, 128, 129, 1000, 2049, 4097, 5000, 4990)[int(rng.integers(0, 8))])
        a = rng.integers(0, 4, m).astype(np.uint8)
        out = []; j = 0; d = (0.02, 0.1, 0.3)[i % 3]
        while j < m and len(out) < n:
            r = rng.random()
            if r < d * 0.6: out.append((int(a[j]) + 1 + int(rng.integers(0, 3))) & 3); j += 1
            elif r < d * 0.8: out.extend(rng.integers(0, 4, int(rng.integers(1, 30))).tolist())
            elif r < d: j += int(rng.integers(1, 30))
            else: out.append(int(a[j])); j += 1
        out = out[:n]
        while len(out) < n: out.append(int(rng.integers(0, 4)))
        A = np.frombuffer(b"ACGT", dtype=np.uint8)[a].copy(); B = np.frombuffer(b"ACGT", dtype=np.uint8)[np.array(out, dtype=np.uint8)].copy()
        s1.append(A.tobytes()); s2.append(B.tobytes())
    return s1, s2


def main():
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle_py.build(ref=False)
    s1, s2 = make_pairs(npairs, seed)
    if os.environ.get("DP_FUZZ_LARGE"):
        l1, l2 = make_large_pairs(int(os.environ["DP_FUZZ_LARGE"]), seed); s1 += l1; s2 += l2; npairs = len(s1)
    g = capi.Aligner.for_leaf_operators() if hasattr(capi.Aligner, "for_leaf_operators") else None
    if g is None:
        import gzip, shutil, tempfile
        from gsalign_amd import indexio
        tmp = tempfile.mkdtemp()
        for ext in ("bwt", "sa", "pac", "ann", "amb"):
            with gzip.open(os.path.join(ROOT, "tests", "golden", f"small.{ext}.gz"), "rb") as a, open(os.path.join(tmp, f"small.{ext}"), "wb") as b:
                shutil.copyfileobj(a, b)
        g = capi.Aligner(indexio.load_index(os.path.join(tmp, "small")))
    bad = 0
    for rep in range(2):      # (twice: the second call reuses buffers, counters and the boundary epoch)
        ops = g.ksw2_batch(s1, s2)
        for i in range(npairs):
            a1, a2 = oracle_py.oracle_ksw2(s1[i], s2[i])
            if capi.apply_ops(s1[i], s2[i], ops[i]) != (a1, a2):
                bad += 1
                if bad <= 10: print(f"MISMATCH rep {rep} pair {i}: m={len(s1[i])} n={len(s2[i])}", flush=True)
    print(f"dp_fuzz: {npairs} pairs x 2 calls, seed {seed}: {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
