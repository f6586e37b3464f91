"""ctypes binding of libgsa_host.so: the CPU-side components (index builder, MAF/VCF
emitters) behind a small C API, so the CPU test-suite can check them without a GPU."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import capi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libgsa_host.so")
CLI_PATH = os.path.join(HERE, "bin", "GSAlign_hip")

RESULT_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_char), C.c_int, C.POINTER(capi.Result))


def build() -> None:
    subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "-j8", "all"], check=True, stdout=subprocess.DEVNULL)


def load() -> C.CDLL:
    path = os.environ.get("GSA_HOST_LIB_PATH") or LIB_PATH      # (an instrumented build of the same sources: tools/host_tsan.sh)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run __graft_entry__.build()")
    return C.CDLL(path)


def build_index(fasta: str, prefix: str) -> None:
    err = C.create_string_buffer(256)
    if load().gsah_c_build_index(fasta.encode(), prefix.encode(), err) != 0:
        raise RuntimeError(err.value.decode())


def result_from_dump(d: dict, keep: list):
    """dict in the oracle/capi 'blocks_as_dump' layout -> a populated capi.Result (arrays appended to `keep`)."""
    nb = d["b_score"].size; nf = d["f_qpos"].size
    B = np.zeros(nb, capi.BLOCK_DT); F = np.zeros(nf, capi.FRAG_DT)
    B["score"] = d["b_score"]; B["aln_len"] = d["b_aln_len"]; B["bdup"] = d["b_bdup"]; B["n_frag"] = d["b_nfrag"]
    B["frag_off"] = np.concatenate([[0], np.cumsum(d["b_nfrag"][:-1], dtype=np.int64)]) if nb else 0
    B["bdir"] = d["b_bdir"]; B["gpos"] = d["b_gpos"]; B["chr"] = d["b_chr"]
    F["bseed"] = d["f_bseed"]; F["qpos"] = d["f_qpos"]; F["qlen"] = d["f_qlen"]; F["rlen"] = d["f_rlen"]; F["rpos"] = d["f_rpos"]
    F["aln_len"] = d["f_alnlen"]
    F["aln_off"] = np.concatenate([[0], np.cumsum(d["f_alnlen"][:-1], dtype=np.int64)]) if nf else 0
    a1 = np.ascontiguousarray(d["aln1"]); a2 = np.ascontiguousarray(d["aln2"])
    keep.extend([B, F, a1, a2])
    r = capi.Result()
    r.n_blocks = nb; r.n_frags = nf; r.n_aln = a1.size
    R = capi.pack_recs(F); keep.append(R)
    r.blocks = C.cast(B.ctypes.data, C.POINTER(capi.Block)); r.recs = C.cast(R.ctypes.data, C.POINTER(capi.Rec))
    r.aln1 = C.cast(a1.ctypes.data, C.POINTER(C.c_char)); r.aln2 = C.cast(a2.ctypes.data, C.POINTER(C.c_char))
    return r


def emit(index_prefix: str, query_fa: str, maf_path: str, vcf_path: str, reference_label: str, per_contig, allow_dup: bool = True, fmt: int = 1) -> None:
    """per_contig(ci, seq_uint8) -> dump dict of the finished contig (stage 8 layout).  fmt 1 = MAF, 2 = ALN (written to maf_path)."""
    keep: list = []

    def cb(user, ci, seq, ln, out):
        s = np.frombuffer(C.string_at(seq, ln), dtype=np.uint8)
        r = result_from_dump(per_contig(ci, s), keep)
        C.memmove(out, C.byref(r), C.sizeof(capi.Result))
        return 0

    err = C.create_string_buffer(256)
    rc = load().gsah_c_emit_fmt(index_prefix.encode(), query_fa.encode(), maf_path.encode(), vcf_path.encode(), reference_label.encode(),
                                1 if allow_dup else 0, fmt, RESULT_CB(cb), None, err)
    if rc != 0:
        raise RuntimeError(f"gsah_c_emit -> {rc}: {err.value.decode()}")


def dotplot(index_prefix: str, query_fa: str, contig: int, gp_path: str, out_prefix: str, per_contig) -> bool:
    """OutputDotplot (DotPloting.cpp:10-71) for one contig: gnuplot script + data files, gnuplot itself is not run."""
    keep: list = []

    def cb(user, ci, seq, ln, out):
        s = np.frombuffer(C.string_at(seq, ln), dtype=np.uint8)
        r = result_from_dump(per_contig(ci, s), keep)
        C.memmove(out, C.byref(r), C.sizeof(capi.Result))
        return 0

    err = C.create_string_buffer(256)
    rc = load().gsah_c_dotplot(index_prefix.encode(), query_fa.encode(), contig, gp_path.encode(), out_prefix.encode(), RESULT_CB(cb), None, err)
    if rc < 0:
        raise RuntimeError(f"gsah_c_dotplot -> {rc}: {err.value.decode()}")
    return rc == 1
