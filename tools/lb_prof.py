#!/usr/bin/env python3
"""Where a tile of a fused look-back pass spends its time (experiment build: make lib VARIANT=lbt EXTRA=-DLB_TIMING; GSA_LIB_PATH=.../libgsa_hip_lbt.so).
The chaining passes (k_chain.hip) of (a) a 60 Mb -sen bundle of yeast-sized contigs, (b) one 60 Mb contig with default parameters, one context alone:
tick sums (100 MHz) of thread 0 of every tile -- ticket draw, loads + own scan, look-back, emit -- divided by the tiles."""
import ctypes as C, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsalign_amd import capi, synth, indexio, hostlib

YEAST_KB = [230, 813, 317, 1532, 577, 270, 1091, 563, 440, 746, 667, 1078, 924, 784, 1091, 948]


def prof(lib, reset=True):
    out = (C.c_uint64 * 8)()
    assert lib.gsa_debug_lb_prof(out, 1 if reset else 0) == 0
    return np.array(list(out), dtype=np.float64)


def report(tag, p, wall_ms):
    tiles = max(p[4], 1.0)
    us = p[:4] / 100.0 / tiles
    print(f"{tag}: {int(p[4])} tiles, wall {wall_ms:.2f} ms; per tile (us): ticket {us[0]:.2f}  loads+scan {us[1]:.2f}  look-back {us[2]:.2f}  emit {us[3]:.2f}  = {us.sum():.2f}")


def main():
    tmp = tempfile.mkdtemp(prefix="lbprof")
    for tag, lengths, div, params, reps in (("yeast -sen bundles", [1000 * k for k in YEAST_KB], 0.02, dict(sen=1, clr=50), 5), ("one 60 Mb contig", [60_000_000], 0.01, {}, 1)):
        refs, qrys = synth.make_pair_fast(0, len(lengths), div, seed=7, lengths=lengths)
        rf, px = os.path.join(tmp, "r.fa"), os.path.join(tmp, "r")
        synth.write_fasta(rf, refs); hostlib.build_index(rf, px)
        idx = indexio.load_index(px)
        g = capi.Aligner(idx, **params)
        lib = g.lib
        lib.gsa_debug_lb_prof.argtypes = [C.POINTER(C.c_uint64), C.c_int]
        contigs = [g.pinned_copy(q) for _, q in qrys] * reps
        for rep in range(3):
            prof(lib)
            t0 = time.perf_counter()
            capi.align_many([g], contigs, None)
            wall = (time.perf_counter() - t0) * 1e3
            report(f"{tag} rep {rep}", prof(lib), wall)
        g.close()


if __name__ == "__main__":
    main()
