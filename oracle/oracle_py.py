"""ctypes access to the two CHECKER libraries (test infrastructure only).

* ``Oracle``  -- oracle/libgsa_oracle.so, our CPU restatement (gsa_oracle.cpp)
* ``RefLib``  -- oracle/_ref/libgsref.so, the real reference objects + ref_glue.cpp

Both expose the same stage-dump getters, so ``dump_stages`` works on either.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.

The reference keeps its state in process globals and can load ONE index per
process, so run ``python oracle/oracle_py.py refdump ...`` in a subprocess (see
``ref_dump_subprocess``) when more than one index is needed.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ORACLE_SO = os.path.join(HERE, "libgsa_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libgsref.so")
REF_GSALIGN = os.path.join(HERE, "_ref", "GSAlign_ref")
REF_BWT_INDEX = os.path.join(HERE, "_ref", "bwt_index_ref")

DEFAULT_PARAMS = dict(slen=15, ind=25, clr=200, alen=200, idy=70, sen=0, one=0)   # main.cpp:202-214


def build(ref: bool = True) -> None:
    """make the oracle (and, when /root/reference is present, oracle/_ref)."""
    subprocess.run(["make", "-C", HERE, "oracle"] + (["ref"] if ref else []), check=True, stdout=subprocess.DEVNULL)
    if ref and os.path.exists(os.path.join(ROOT, "gsalign_amd", "lib", "libgsa_hip.so")):
        subprocess.run(["make", "-C", HERE, "ref_hip"], check=True, stdout=subprocess.DEVNULL)      # INTEGRATION.md section 2, compiled (needs the product library)


def have_ref() -> bool:
    return os.path.exists(REF_SO) and os.path.exists(REF_GSALIGN) and os.path.exists(REF_BWT_INDEX)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class _StageReader:
    """Shared getters; subclasses provide self._f(name) -> bound C function taking the right ctx."""

    def _call(self, name, *args):
        raise NotImplementedError

    def seeds(self):
        n = self._call("seed_count")
        q = np.empty(n, np.int32); l = np.empty(n, np.int32); r = np.empty(n, np.int64)
        self._call("seeds", _p(q, C.c_int), _p(l, C.c_int), _p(r, C.c_longlong))
        return q, l, r

    def groups(self):
        n = self._call("group_count")
        b = np.empty(n, np.int32); e = np.empty(n, np.int32)
        self._call("groups", _p(b, C.c_int), _p(e, C.c_int))
        return b, e

    def blocks(self, with_aln: bool = False):
        nb = self._call("block_count")
        m = {k: np.empty(nb, np.int32) for k in ("score", "aln_len", "bdup", "nfrag", "bdir", "gpos", "chr")}
        self._call("block_meta", *[_p(m[k], C.c_int) for k in ("score", "aln_len", "bdup", "nfrag", "bdir", "gpos", "chr")])
        nf = self._call("frag_total")
        f = {k: np.empty(nf, np.int32) for k in ("bseed", "qpos", "qlen", "rlen", "alnlen")}
        f["rpos"] = np.empty(nf, np.int64)
        self._call("frags", _p(f["bseed"], C.c_int), _p(f["qpos"], C.c_int), _p(f["qlen"], C.c_int),
                   _p(f["rpos"], C.c_longlong), _p(f["rlen"], C.c_int), _p(f["alnlen"], C.c_int))
        out = {"b_" + k: v for k, v in m.items()}
        out.update({"f_" + k: v for k, v in f.items()})
        if with_aln:
            na = self._call("aln_total")
            a1 = np.zeros(max(na, 1), np.uint8); a2 = np.zeros(max(na, 1), np.uint8)
            self._call("frag_aln", _p(a1, C.c_char), _p(a2, C.c_char))
            out["aln1"] = a1[:na]; out["aln2"] = a2[:na]
        return out

    def dump_stages(self, upto: int = 8) -> dict:
        """Run stage by stage and collect everything the parity tests compare."""
        d = {}
        for st in range(1, upto + 1):
            self.run_to(st)
            if st == 1:
                q, l, r = self.seeds(); b, e = self.groups()
                d.update(s1_qpos=q, s1_qlen=l, s1_rpos=r, s1_gbeg=b, s1_gend=e)
            else:
                for k, v in self.blocks(with_aln=(st == 8)).items():
                    d[f"s{st}_{k}"] = v
        return d


class Oracle(_StageReader):
    def __init__(self, idx, params: dict | None = None):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        self.lib = C.CDLL(ORACLE_SO)
        L = self.lib
        L.ora_create.restype = C.c_void_p
        L.ora_create.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64,
                                 C.c_char_p, C.c_int64, C.POINTER(C.c_int32), C.c_int]
        for n in ("seed_count", "frag_total", "aln_total", "bwt_sa"):
            getattr(L, "ora_" + n).restype = C.c_longlong
        self.idx = idx
        ref = np.ascontiguousarray(idx.ref)
        self.ctx = C.c_void_p(L.ora_create(_p(idx.hdr, C.c_uint64), _p(idx.bwt, C.c_uint32), idx.bwt.size,
                                           _p(idx.sa, C.c_uint64), idx.sa.size, ref.ctypes.data_as(C.c_char_p),
                                           idx.G, _p(idx.chr_len, C.c_int32), len(idx.chr_len)))
        self.set_params(**(params or {}))

    def _call(self, name, *args):
        return getattr(self.lib, "ora_" + name)(self.ctx, *args)

    def set_params(self, **kw):
        p = dict(DEFAULT_PARAMS); p.update(kw)
        self.lib.ora_params(self.ctx, p["slen"], p["ind"], p["clr"], p["alen"], p["idy"], p["sen"], p["one"])

    def set_query(self, seq: np.ndarray):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self._q = seq
        self.lib.ora_set_query(self.ctx, seq.ctypes.data_as(C.c_char_p), seq.size)

    def run_to(self, stage: int):
        return self.lib.ora_run_to(self.ctx, stage)

    def counters(self):
        c = np.zeros(8, np.uint64)
        self.lib.ora_counters(self.ctx, _p(c, C.c_uint64))
        return c

    def gap_similarity(self, q1, q2, r1, r2):
        return self.lib.ora_gap_similarity(self.ctx, q1, q2, C.c_longlong(r1), C.c_longlong(r2))

    def bwt_search(self, start, stop):
        ln = C.c_int(0); locs = np.zeros(100, np.int64)
        f = self.lib.ora_bwt_search(self.ctx, start, stop, C.byref(ln), _p(locs, C.c_longlong))
        return ln.value, locs[:f].copy()

    def ksw2(self, s1: bytes, s2: bytes):
        o1 = C.create_string_buffer(len(s1) + len(s2) + 1); o2 = C.create_string_buffer(len(s1) + len(s2) + 1)
        n = self.lib.ora_ksw2(s1, len(s1), s2, len(s2), o1, o2)
        return o1.raw[:n], o2.raw[:n]

    def ksw2_ops(self, s1: bytes, s2: bytes):
        o = C.create_string_buffer(len(s1) + len(s2) + 1)
        n = self.lib.ora_ksw2_ops(s1, len(s1), s2, len(s2), o)
        return o.raw[:n]

    def close(self):
        if self.ctx:
            self.lib.ora_destroy(self.ctx); self.ctx = None


def oracle_ksw2(s1: bytes, s2: bytes):
    """ksw2 restatement without an index."""
    if not os.path.exists(ORACLE_SO):
        build(ref=False)
    lib = C.CDLL(ORACLE_SO)
    o1 = C.create_string_buffer(len(s1) + len(s2) + 1); o2 = C.create_string_buffer(len(s1) + len(s2) + 1)
    n = lib.ora_ksw2(s1, len(s1), s2, len(s2), o1, o2)
    return o1.raw[:n], o2.raw[:n]


class RefLib(_StageReader):
    """The real reference.  ONE index per process."""

    def __init__(self, prefix: str | None, params: dict | None = None):
        self.lib = C.CDLL(REF_SO)
        for n in ("seed_count", "frag_total", "aln_total", "bwt_sa", "genome_size"):
            getattr(self.lib, "gsref_" + n).restype = C.c_longlong
        if prefix is not None:
            rc = self.lib.gsref_init(prefix.encode())
            if rc != 0:
                raise RuntimeError(f"gsref_init({prefix}) -> {rc}")
        self.set_params(**(params or {}))

    def _call(self, name, *args):
        return getattr(self.lib, "gsref_" + name)(*args)

    def set_params(self, **kw):
        p = dict(DEFAULT_PARAMS); p.update(kw)
        self.lib.gsref_params(p["slen"], p["ind"], p["clr"], p["alen"], p["idy"], p["sen"], p["one"])

    def set_query(self, seq: np.ndarray, name: str = "q"):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self.lib.gsref_set_query(name.encode(), seq.ctypes.data_as(C.c_char_p), seq.size)

    def run_to(self, stage: int):
        return self.lib.gsref_run_to(stage)

    def gap_similarity(self, q1, q2, r1, r2):
        return self.lib.gsref_gap_similarity(q1, q2, C.c_longlong(r1), C.c_longlong(r2))

    def bwt_search(self, start, stop):
        ln = C.c_int(0); locs = np.zeros(100, np.int64)
        f = self.lib.gsref_bwt_search(start, stop, C.byref(ln), _p(locs, C.c_longlong))
        return ln.value, locs[:f].copy()

    def ksw2(self, s1: bytes, s2: bytes):
        o1 = C.create_string_buffer(len(s1) + len(s2) + 1); o2 = C.create_string_buffer(len(s1) + len(s2) + 1)
        n = self.lib.gsref_ksw2(s1, len(s1), s2, len(s2), o1, o2)
        return o1.raw[:n], o2.raw[:n]


def ref_build_index(fasta: str, prefix: str) -> None:
    subprocess.run([REF_BWT_INDEX, fasta, prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def ref_run_cli(prefix: str, query_fa: str, out_prefix: str, extra: list | None = None, threads: int = 1) -> None:
    """Run the unmodified reference CLI -> out_prefix.maf / .vcf."""
    cmd = [REF_GSALIGN, "-i", prefix, "-q", query_fa, "-o", out_prefix, "-t", str(threads)] + list(extra or [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def ref_dump_subprocess(prefix: str, query_fa: str, out_npz: str, params: dict | None = None, upto: int = 8, stages=None, wait: bool = True):
    """Stage dumps of the real reference for every contig of query_fa, in a fresh process.  stages: keep only these stages' dumps;
    wait=False: returns the running Popen (several reference processes side by side: one index each, one thread each)."""
    p = dict(DEFAULT_PARAMS); p.update(params or {})
    args = [sys.executable, os.path.abspath(__file__), "refdump", prefix, query_fa, out_npz, str(upto)] + [f"{k}={v}" for k, v in p.items()]
    if stages is not None:
        args.append("stages=" + ",".join(str(int(x)) for x in stages))
    if not wait:
        return subprocess.Popen(args)
    subprocess.run(args, check=True)
    return None


def _main_refdump(argv):
    sys.path.insert(0, ROOT)
    from gsalign_amd.synth import read_fasta
    prefix, query_fa, out_npz, upto = argv[0], argv[1], argv[2], int(argv[3])
    # (stages=1,8: keep only these stages' dumps -- whole genomes: the intermediate block lists are the bulk of the bytes)
    keep = None
    rest = []
    for kv in argv[4:]:
        if kv.startswith("stages="):
            keep = {int(x) for x in kv.split("=")[1].split(",")}
        else:
            rest.append(kv)
    params = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in rest}
    ref = RefLib(prefix, params)
    out = {}
    for ci, (name, seq) in enumerate(read_fasta(query_fa)):
        ref.set_query(seq, name)
        for k, v in ref.dump_stages(upto).items():
            if keep is None or int(k[1:k.index("_")]) in keep:
                out[f"c{ci}_{k}"] = v
    # (chromosome-sized contigs: hundreds of MB of dumps -- zlib would cost more than the stages themselves)
    (np.savez if sum(v.nbytes for v in out.values()) > (64 << 20) else np.savez_compressed)(out_npz, **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "refdump":
        _main_refdump(sys.argv[2:])
    else:
        print("usage: oracle_py.py refdump <index_prefix> <query.fa> <out.npz> <upto> [k=v ...]")
