import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gsalign_amd import capi, indexio, synth
import gzip, shutil, tempfile
tmp = tempfile.mkdtemp()
for ext in ("bwt", "sa", "pac", "ann", "amb"):
    with gzip.open(os.path.join(ROOT, "tests", "golden", f"small.{ext}.gz"), "rb") as a, open(os.path.join(tmp, f"small.{ext}"), "wb") as b:
        shutil.copyfileobj(a, b)
g = capi.Aligner(indexio.load_index(os.path.join(tmp, "small")))
rng = np.random.default_rng(5)
base = synth.fast_genome(4_000_000, 3).tobytes()
for (cnt, lo, hi) in ((190000, 8, 40), (170000, 1, 6), (190000, 30, 60)):
    s1, s2 = [], []
    L = rng.integers(lo, hi + 1, cnt); off = rng.integers(0, 3_900_000, cnt)
    for i in range(cnt):
        a = base[off[i]:off[i] + L[i]]; b = bytearray(a)
        if L[i] > 2: b[L[i] // 2] = 65 if b[L[i] // 2] != 65 else 67
        s1.append(a); s2.append(bytes(b))
    cells = sum(len(x) * len(y) for x, y in zip(s1, s2))
    g.ksw2_batch(s1, s2)
    t = time.time(); g.ksw2_batch(s1, s2); dt = time.time() - t
    print(f"{cnt} pairs {lo}..{hi}: {cells / 1e6:.1f} Mcells, call {dt * 1e3:.2f} ms (incl. copies)", flush=True)
