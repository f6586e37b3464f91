#!/usr/bin/env python3
"""Probe k_dp_stripe: one job per launch over a list of (m, n) shapes, so a
rocprofv3 --kernel-trace of this script gives the per-shape kernel duration.

    rocprofv3 --kernel-trace -d gpurun_out/dpp -o dpp -- python tools/dp_probe.py
    python tools/dp_probe.py --report gpurun_out/dpp/dpp_results.db
"""
import gzip, os, shutil, sqlite3, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(1541, 64), (3000, 64), (64, 1569), (64, 3136), (1541, 1569), (800, 800), (3000, 3000), (200, 200), (400, 100), (100, 400)]
REP = 3


def run():
    import numpy as np
    from gsalign_amd import capi, indexio, synth
    tmp = tempfile.mkdtemp(prefix="gsa_dpp_")
    gold = os.path.join(ROOT, "tests", "golden")
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        with gzip.open(os.path.join(gold, f"small.{ext}.gz"), "rb") as a, open(os.path.join(tmp, f"small.{ext}"), "wb") as b:
            shutil.copyfileobj(a, b)
    gpu = capi.Aligner(indexio.load_index(os.path.join(tmp, "small")), device=0)
    rng = np.random.default_rng(5)
    for m, n in SHAPES:
        s1 = synth.random_genome(m, rng)
        s2 = synth.mutate(synth.random_genome(n, rng) if abs(m - n) > 100 else s1, 0.05, rng)
        s2 = (s2.tobytes() + synth.random_genome(n, rng).tobytes())[:n]
        for _ in range(REP):
            gpu.ksw2_batch([s1.tobytes()], [s2])
    gpu.close(); shutil.rmtree(tmp, ignore_errors=True)


def report(path):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(cur.execute(f"select d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%k_dp_stripe%' order by d.start"))
    for i, (m, n) in enumerate(SHAPES):
        d = [r[0] / 1e3 for r in rows[i * REP:(i + 1) * REP]]
        if d:
            print(f"m={m:5d} n={n:5d} stripes={(n + 63) // 64:3d}  us: " + " ".join(f"{x:8.1f}" for x in d))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        run()
