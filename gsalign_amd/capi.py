"""ctypes binding of libgsa_hip.so (include/gsa_hip.h) for tests and bench.py.

This is plumbing only: it fills `gsa_index_view` from numpy arrays and calls the
C ABI.  There is no CPU path: if the library or a GPU is missing, loading /
`Aligner()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libgsa_hip.so")

EXPORTS = [
    "gsa_default_params", "gsa_create", "gsa_create_opts", "gsa_reserve_index", "gsa_release_reserved", "gsa_clone", "gsa_clone_to_device", "gsa_host_alloc", "gsa_host_free", "gsa_destroy", "gsa_set_params", "gsa_last_error", "gsa_align_contig", "gsa_align_many", "gsa_align_bundle",
    "gsa_align_contig_device", "gsa_set_query_device", "gsa_device_alloc", "gsa_device_free", "gsa_device_upload", "gsa_get_seed_stats", "gsa_hit_buffers", "gsa_seed_chunks", "gsa_hit_count", "gsa_export_hits", "gsa_import_hits", "gsa_finish_contig",
    "gsa_set_query", "gsa_rewind", "gsa_run_to", "gsa_seed_count", "gsa_get_seeds", "gsa_group_count", "gsa_get_groups", "gsa_get_blocks",
    "gsa_bwt_search_batch", "gsa_ksw2_batch", "gsa_gap_similarity_batch", "gsa_get_counters", "gsa_get_timings", "gsa_set_profiling", "gsa_bind_host_thread",
    "gsa_prefetch_contig", "gsa_prefetch_bundle", "gsa_cancel_prefetch", "gsa_get_wall_sums", "gsa_get_alloc_stats", "gsa_debug_buffers", "gsa_set_option", "gsa_host_register", "gsa_host_unregister",
]


class IndexView(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("bwt", C.POINTER(C.c_uint32)), ("bwt_words", C.c_uint64),
                ("sa", C.POINTER(C.c_uint64)), ("n_sa", C.c_uint64), ("ref", C.c_char_p), ("G", C.c_int64),
                ("chr_len", C.POINTER(C.c_int32)), ("n_chr", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("min_seed_len", C.c_int32), ("max_indel", C.c_int32), ("min_block_score", C.c_int32), ("min_aln_len", C.c_int32),
                ("min_identity", C.c_int32), ("sensitive", C.c_int32), ("one_on_one", C.c_int32)]


class Seed(C.Structure):
    _fields_ = [("qpos", C.c_int32), ("len", C.c_int32), ("rpos", C.c_int64)]


class Frag(C.Structure):
    _fields_ = [("bseed", C.c_int32), ("qpos", C.c_int32), ("qlen", C.c_int32), ("rlen", C.c_int32), ("rpos", C.c_int64),
                ("aln_off", C.c_int64), ("aln_len", C.c_int32), ("_pad", C.c_int32)]


class Block(C.Structure):
    _fields_ = [("score", C.c_int32), ("aln_len", C.c_int32), ("bdup", C.c_int32), ("n_frag", C.c_int32), ("frag_off", C.c_int64),
                ("bdir", C.c_int32), ("gpos", C.c_int32), ("chr", C.c_int32), ("_pad", C.c_int32)]


class Rec(C.Structure):
    """gsa_rec, 16 bytes: seed {qpos >= 0, len, rpos} or gap {-1 - qlen, rlen, aln_len, aln_off} (include/gsa_hip.h)."""
    _fields_ = [("w0", C.c_int32), ("w1", C.c_int32), ("w2", C.c_int32), ("w3", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("n_frags", C.c_int64), ("n_aln", C.c_int64), ("blocks", C.POINTER(Block)),
                ("recs", C.POINTER(Rec)), ("aln1", C.POINTER(C.c_char)), ("aln2", C.POINTER(C.c_char))]


FRAG_DT = np.dtype([("bseed", "<i4"), ("qpos", "<i4"), ("qlen", "<i4"), ("rlen", "<i4"), ("rpos", "<i8"), ("aln_off", "<i8"), ("aln_len", "<i4"), ("_pad", "<i4")])
BLOCK_DT = np.dtype([("score", "<i4"), ("aln_len", "<i4"), ("bdup", "<i4"), ("n_frag", "<i4"), ("frag_off", "<i8"), ("bdir", "<i4"), ("gpos", "<i4"), ("chr", "<i4"), ("_pad", "<i4")])
SEED_DT = np.dtype([("qpos", "<i4"), ("len", "<i4"), ("rpos", "<i8")])
REC_DT = np.dtype([("w0", "<i4"), ("w1", "<i4"), ("w2", "<i4"), ("w3", "<u4")])


def expand_recs(recs: np.ndarray) -> np.ndarray:
    """gsa_rec[n] -> gsa_frag[n] (FRAG_DT): numpy form of gsa_expand_frags (include/gsa_hip.h)."""
    n = recs.size
    F = np.zeros(n, FRAG_DT)
    if not n:
        return F
    seed = recs["w0"] >= 0
    rpos = recs.view(np.int64).reshape(-1, 2)[:, 1]
    F["bseed"] = seed
    F["qpos"][seed] = recs["w0"][seed]; F["qlen"][seed] = recs["w1"][seed]; F["rlen"][seed] = recs["w1"][seed]; F["rpos"][seed] = rpos[seed]
    g = np.flatnonzero(~seed)
    if g.size:
        assert g[0] > 0 and seed[g - 1].all(), "a gap record follows a seed record"
        F["qpos"][g] = recs["w0"][g - 1] + recs["w1"][g - 1]; F["rpos"][g] = rpos[g - 1] + recs["w1"][g - 1]
        F["qlen"][g] = -1 - recs["w0"][g]; F["rlen"][g] = recs["w1"][g]; F["aln_len"][g] = recs["w2"][g]; F["aln_off"][g] = recs["w3"][g]
    return F


def pack_recs(F: np.ndarray) -> np.ndarray:
    """gsa_frag[n] -> gsa_rec[n]: the inverse (what the library's last pass does on the device)."""
    R = np.zeros(F.size, REC_DT)
    seed = F["bseed"] != 0
    rpos = R.view(np.int64).reshape(-1, 2)[:, 1]
    R["w0"][seed] = F["qpos"][seed]; R["w1"][seed] = F["qlen"][seed]; rpos[seed] = F["rpos"][seed]
    g = ~seed
    R["w0"][g] = -1 - F["qlen"][g]; R["w1"][g] = F["rlen"][g]; R["w2"][g] = F["aln_len"][g]; R["w3"][g] = F["aln_off"][g].astype(np.uint32)
    return R


def build_library() -> None:
    """hipcc cross-compiles for gfx950 without a GPU (seconds per file once warm)."""
    subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "-j8", "lib"], check=True, stdout=subprocess.DEVNULL)


def load_library() -> C.CDLL:
    path = os.environ.get("GSA_LIB_PATH") or LIB_PATH          # (GSA_LIB_PATH: an experiment variant of the library, tools only)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)")
    lib = C.CDLL(path)
    lib.gsa_create.argtypes = [C.c_int, C.POINTER(IndexView), C.POINTER(Params), C.POINTER(C.c_void_p)]
    lib.gsa_create_opts.argtypes = [C.c_int, C.POINTER(IndexView), C.POINTER(Params), C.c_uint32, C.POINTER(C.c_void_p)]
    lib.gsa_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.gsa_clone_to_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.gsa_host_alloc.restype = C.c_void_p
    lib.gsa_host_alloc.argtypes = [C.c_size_t]
    lib.gsa_host_free.argtypes = [C.c_void_p]
    lib.gsa_device_alloc.restype = C.c_void_p
    lib.gsa_device_alloc.argtypes = [C.c_int, C.c_size_t]
    lib.gsa_device_free.argtypes = [C.c_int, C.c_void_p]
    lib.gsa_device_upload.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.gsa_align_contig_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(Result)]
    lib.gsa_set_query_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.gsa_prefetch_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    lib.gsa_cancel_prefetch.argtypes = [C.c_void_p]
    lib.gsa_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.gsa_last_error.restype = C.c_char_p
    lib.gsa_last_error.argtypes = [C.c_void_p]
    lib.gsa_seed_count.restype = C.c_int64
    lib.gsa_hit_count.restype = C.c_int64
    lib.gsa_hit_count.argtypes = [C.c_void_p]
    lib.gsa_export_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gsa_import_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    for name in ("gsa_destroy", "gsa_set_params", "gsa_align_contig", "gsa_set_query", "gsa_run_to", "gsa_seed_count", "gsa_get_seeds",
                 "gsa_group_count", "gsa_get_groups", "gsa_get_blocks", "gsa_bwt_search_batch", "gsa_ksw2_batch", "gsa_gap_similarity_batch",
                 "gsa_get_counters", "gsa_get_timings", "gsa_set_profiling"):
        fn = getattr(lib, name)
        if fn.argtypes is None:
            fn.argtypes = None
    return lib


def bind_host_thread(device: int = 0) -> None:
    """gsa_bind_host_thread: the calling thread moves to the CPUs next to `device` (threads started later inherit that)."""
    lib = load_library()
    lib.gsa_bind_host_thread.argtypes = [C.c_int]
    lib.gsa_bind_host_thread(int(device))


RESULT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(Result))


class DeviceContig:
    """A query contig resident in device memory (gsa_device_alloc + gsa_device_upload): what gsa_align_contig_device takes."""

    def __init__(self, lib, device: int, seq: np.ndarray):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self.lib, self.device, self.size = lib, int(device), int(seq.size)
        self.ptr = lib.gsa_device_alloc(self.device, self.size)
        if not self.ptr:
            raise GsaError("gsa_device_alloc failed")
        if lib.gsa_device_upload(self.device, C.c_void_p(self.ptr), C.c_void_p(seq.ctypes.data), self.size) != 0:
            raise GsaError("gsa_device_upload failed")

    def free(self):
        if self.ptr:
            self.lib.gsa_device_free(self.device, C.c_void_p(self.ptr)); self.ptr = None


def align_many(aligners, contigs, on_result=None, in_order: bool = False, bundle: bool = True, prefetch: bool = True) -> None:
    """gsa_align_many: `contigs` (uint8 arrays, or DeviceContig objects -- all of one kind) on the given contexts, one host
    thread per context inside the library.  on_result(contig_index, Result) runs on the worker threads (the Result is valid
    during the call only).  in_order: hand the contigs out as listed (GSA_MANY_IN_ORDER) instead of longest first.
    bundle=False: GSA_MANY_NO_BUNDLE (every contig in a pass of its own; by default short contigs share passes).
    prefetch=False: GSA_MANY_NO_PREFETCH (a contig is uploaded when its turn comes; by default a context uploads its next contig
    while it aligns the current one)."""
    lib = aligners[0].lib
    n = len(contigs)
    ctxs = (C.c_void_p * len(aligners))(*[a.ctx for a in aligners])
    on_dev = n > 0 and isinstance(contigs[0], DeviceContig)
    qs = (C.c_char_p * n)(*[C.cast(c.ptr if on_dev else c.ctypes.data, C.c_char_p) for c in contigs])
    ql = (C.c_int32 * n)(*[int(c.size) for c in contigs])
    cb = RESULT_FN((lambda user, ci, res: int(on_result(ci, res.contents) or 0)) if on_result else 0)
    lib.gsa_align_many.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32, C.c_uint32, RESULT_FN, C.c_void_p]
    rc = lib.gsa_align_many(ctxs, len(aligners), qs, ql, n, (1 if in_order else 0) | (2 if on_dev else 0) | (0 if bundle else 8) | (0 if prefetch else 16), cb, None)
    if rc != 0:
        msgs = [lib.gsa_last_error(a.ctx).decode() for a in aligners]
        raise GsaError(f"gsa_align_many -> {rc}: {'; '.join(m for m in msgs if m)}")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def result_as_dump(r: dict, with_aln: bool = False) -> dict:
    """A result dict (Aligner._result: blocks / frags / aln1 / aln2) in the shape of oracle_py._StageReader.blocks(): records
    gathered in block order."""
    B, F = r["blocks"], r["frags"]
    idx = np.concatenate([np.arange(o, o + n) for o, n in zip(B["frag_off"], B["n_frag"])]) if B.size else np.zeros(0, np.int64)
    Fo = F[idx] if idx.size else np.zeros(0, FRAG_DT)
    out = {"b_score": B["score"].copy(), "b_aln_len": B["aln_len"].copy(), "b_bdup": B["bdup"].copy(), "b_nfrag": B["n_frag"].copy(),
           "b_bdir": B["bdir"].copy(), "b_gpos": B["gpos"].copy(), "b_chr": B["chr"].copy(),
           "f_bseed": Fo["bseed"].copy(), "f_qpos": Fo["qpos"].copy(), "f_qlen": Fo["qlen"].copy(), "f_rlen": Fo["rlen"].copy(),
           "f_alnlen": Fo["aln_len"].copy(), "f_rpos": Fo["rpos"].copy()}
    if with_aln:
        segs1 = [r["aln1"][o:o + n] for o, n in zip(Fo["aln_off"], Fo["aln_len"]) if n]
        segs2 = [r["aln2"][o:o + n] for o, n in zip(Fo["aln_off"], Fo["aln_len"]) if n]
        out["aln1"] = np.concatenate(segs1) if segs1 else np.zeros(0, np.uint8)
        out["aln2"] = np.concatenate(segs2) if segs2 else np.zeros(0, np.uint8)
    return out


class GsaError(RuntimeError):
    pass


class Aligner:
    """One gsa_ctx on one GPU."""

    # gsa_set_option names a test / tool may also give through the environment (GSA_<NAME>): the LIBRARY reads no environment variable,
    # this Python layer does and passes the values on
    ENV_OPTIONS = ("split_min", "bundle_contig", "bundle_cap", "seed_budget", "dp_lane", "seed_mode", "pd_bitmap", "walk_chain_min", "pd_two_level_min", "pres_from_kmer", "pd_bytes", "dp_occupancy", "sweep_shape", "dp_side", "dp_small_side", "dp_safe", "dp_fake_timeout")

    def _options_from_env(self):
        for name in self.ENV_OPTIONS:
            v = os.environ.get("GSA_" + name.upper())
            if v is not None and v != "":
                self.set_option(name, {"sweep": 0, "spec": 1, "search": 2}.get(v, v) if name == "seed_mode" else v)
        if os.environ.get("GSA_NO_PDBITMAP"):
            self.set_option("pd_bitmap", 0)

    def set_option(self, name: str, value) -> None:
        self._ck(self.lib.gsa_set_option(self.ctx, name.encode(), C.c_int64(int(value))))

    def __init__(self, idx, device: int = 0, wide: bool = False, kmer_k: int = 0, pac=None, _clone_of=None, _copy_to=None, **params):
        self.lib = load_library()
        self.idx = idx
        self._pinned = []
        self._devbufs = []
        if _clone_of is not None and _copy_to is not None:
            self.ctx = C.c_void_p()
            rc = self.lib.gsa_clone_to_device(_clone_of.ctx, int(_copy_to), C.byref(self.ctx))
            if rc != 0:
                raise GsaError(f"gsa_clone_to_device -> {rc}: {self.lib.gsa_last_error(None).decode()}")
            self._options_from_env()          # (independent of its parent: owns its copy of the index)
            return
        if _clone_of is not None:
            self.ctx = C.c_void_p()
            rc = self.lib.gsa_clone(_clone_of.ctx, C.byref(self.ctx))
            if rc != 0:
                raise GsaError(f"gsa_clone -> {rc}: {self.lib.gsa_last_error(None).decode()}")
            self._parent = _clone_of        # keeps the index owner alive
            self._options_from_env()
            return
        self._ref = np.ascontiguousarray(idx.ref)
        v = IndexView()
        v.primary = int(idx.hdr[0])
        v.L2[0] = 0
        for i in range(1, 5):
            v.L2[i] = int(idx.hdr[i])
        v.bwt = _p(idx.bwt, C.c_uint32); v.bwt_words = idx.bwt.size
        v.sa = _p(idx.sa, C.c_uint64); v.n_sa = idx.sa.size
        v.ref = self._ref.ctypes.data_as(C.c_char_p); v.G = idx.G
        if pac is not None:      # GSA_CREATE_REF_PAC: the bytes of the .pac file instead of RefSequence -- the device unpacks the text itself
            self._pac = np.ascontiguousarray(pac, dtype=np.uint8)
            assert self._pac.size >= (idx.G + 3) // 4
            v.ref = self._pac.ctypes.data_as(C.c_char_p)
        v.chr_len = _p(idx.chr_len, C.c_int32); v.n_chr = len(idx.chr_len)
        self.ctx = C.c_void_p()
        p = self._params(**params)
        # wide=True (or GSA_FORCE_WIDE=1 in the environment of the test run): GSA_CREATE_WIDE, the >= 2^32-row device layout on any index;
        # kmer_k (or GSA_KMER_K): GSA_CREATE_KMER_K
        wide = wide or os.environ.get("GSA_FORCE_WIDE", "0") not in ("", "0")
        kmer_k = kmer_k or int(os.environ.get("GSA_KMER_K", "0") or 0)
        prio = int(os.environ.get("GSA_PRIO", "0") or 0)          # GSA_CREATE_PRIO (experiments / bench: stream priorities)
        flags = (1 if wide else 0) | ((kmer_k & 15) << 8) | ((prio & 3) << 16) | (4 if pac is not None else 0)
        rc = self.lib.gsa_create_opts(device, C.byref(v), C.byref(p), flags, C.byref(self.ctx))
        if rc != 0:
            raise GsaError(f"gsa_create -> {rc}: {self.lib.gsa_last_error(None).decode()}")
        self._options_from_env()

    def clone(self) -> "Aligner":
        """A further context on the same GPU sharing this one's device index (gsa_clone)."""
        return Aligner(self.idx, _clone_of=self)

    def clone_to_device(self, device: int) -> "Aligner":
        """A context on GPU `device` whose device index is a device-to-device COPY of this one's (gsa_clone_to_device): no upload, no table builds."""
        return Aligner(self.idx, _clone_of=self, _copy_to=device)

    def pinned_copy(self, seq: np.ndarray) -> np.ndarray:
        """seq copied into pinned host memory from gsa_host_alloc (what a FASTA loader of an integrated host reads into)."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        p = self.lib.gsa_host_alloc(seq.size + 64)
        if not p:
            raise GsaError("gsa_host_alloc failed")
        self._pinned.append(p)
        buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(seq.size,))
        buf[:] = seq
        return buf

    def device_copy(self, seq: np.ndarray, device: int = 0) -> DeviceContig:
        """seq uploaded once into device memory (gsa_device_alloc / gsa_device_upload); freed by close()."""
        d = DeviceContig(self.lib, device, seq)
        self._devbufs.append(d)
        return d

    def prefetch_contig(self, seq: np.ndarray):
        """gsa_prefetch_contig: start the upload of the NEXT contig; the following align_contig(seq) -- same buffer -- finds it on the device."""
        self._ck(self.lib.gsa_prefetch_contig(self.ctx, seq.ctypes.data_as(C.c_char_p), C.c_int32(seq.size)))

    def cancel_prefetch(self):
        self._ck(self.lib.gsa_cancel_prefetch(self.ctx))

    def align_contig_device(self, d: DeviceContig) -> dict:
        res = Result()
        self._ck(self.lib.gsa_align_contig_device(self.ctx, C.c_void_p(d.ptr), C.c_int32(d.size), C.byref(res)))
        return self._result(res)

    def seed_stats(self) -> np.ndarray:
        st = np.zeros(8, np.uint64)
        self._ck(self.lib.gsa_get_seed_stats(self.ctx, _p(st, C.c_uint64)))
        return st

    def _params(self, slen=15, ind=25, clr=200, alen=200, idy=70, sen=0, one=0) -> Params:
        return Params(slen, ind, clr, alen, idy, sen, one)

    def _ck(self, rc):
        if rc != 0:
            raise GsaError(f"libgsa_hip error {rc}: {self.lib.gsa_last_error(self.ctx).decode()}")

    def set_params(self, **params):
        p = self._params(**params)
        self._ck(self.lib.gsa_set_params(self.ctx, C.byref(p)))

    def set_profiling(self, on: bool, count_blocks: bool = False, seed_only: bool = False):
        self._ck(self.lib.gsa_set_profiling(self.ctx, (1 if on else 0) | (2 if count_blocks else 0) | (4 if seed_only else 0)))

    def set_query(self, seq: np.ndarray):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self._q = seq
        self._ck(self.lib.gsa_set_query(self.ctx, seq.ctypes.data_as(C.c_char_p), C.c_int32(seq.size)))

    def rewind(self):
        """Stage 0 again with the uploaded contig (no new copy)."""
        self._ck(self.lib.gsa_rewind(self.ctx))

    def run_to(self, stage: int):
        self._ck(self.lib.gsa_run_to(self.ctx, stage))

    def align_contig(self, seq: np.ndarray) -> dict:
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self._q = seq
        res = Result()
        self._ck(self.lib.gsa_align_contig(self.ctx, seq.ctypes.data_as(C.c_char_p), C.c_int32(seq.size), C.byref(res)))
        return self._result(res)

    # ---- one contig seeded by several GPUs (gsa_seed_chunks ... gsa_finish_contig) ----
    def seed_chunks(self, seq: np.ndarray, chunk_beg: int, chunk_end: int) -> int:
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self._q = seq
        self._ck(self.lib.gsa_seed_chunks(self.ctx, seq.ctypes.data_as(C.c_char_p), C.c_int32(seq.size), C.c_int32(chunk_beg), C.c_int32(chunk_end)))
        return int(self.lib.gsa_hit_count(self.ctx))

    def hit_count(self) -> int:
        return int(self.lib.gsa_hit_count(self.ctx))

    def export_hits(self, keys_ptr=None, vals_ptr=None):
        """Hits of this context's chunk range.  Without arguments: two numpy arrays; with pointers (host or device memory of
        gsa_hit_count entries): copied there."""
        n = int(self.lib.gsa_hit_count(self.ctx))
        if keys_ptr is not None:
            self._ck(self.lib.gsa_export_hits(self.ctx, C.c_void_p(keys_ptr), C.c_void_p(vals_ptr)))
            return n
        k = np.zeros(n, np.uint64); v = np.zeros(n, np.uint32)
        if n:
            self._ck(self.lib.gsa_export_hits(self.ctx, C.c_void_p(k.ctypes.data), C.c_void_p(v.ctypes.data)))
        return k, v

    def import_hits(self, keys, vals, n=None):
        """keys / vals: numpy arrays, or raw pointers (host or device memory) with n."""
        if n is None:
            keys = np.ascontiguousarray(keys, np.uint64); vals = np.ascontiguousarray(vals, np.uint32)
            n = int(keys.size); kp, vp = keys.ctypes.data, vals.ctypes.data
        else:
            kp, vp = keys, vals
        self._ck(self.lib.gsa_import_hits(self.ctx, C.c_void_p(kp), C.c_void_p(vp), C.c_int64(n)))

    def finish_contig(self) -> dict:
        res = Result()
        self._ck(self.lib.gsa_finish_contig(self.ctx, C.byref(res)))
        return self._result(res)

    def seeds(self):
        n = self.lib.gsa_seed_count(self.ctx)
        out = np.zeros(n, dtype=SEED_DT)
        if n:
            self._ck(self.lib.gsa_get_seeds(self.ctx, out.ctypes.data_as(C.c_void_p)))
        return out["qpos"].copy(), out["len"].copy(), out["rpos"].copy()

    def groups(self):
        n = self.lib.gsa_group_count(self.ctx)
        b = np.zeros(n, np.int32); e = np.zeros(n, np.int32)
        if n:
            self._ck(self.lib.gsa_get_groups(self.ctx, _p(b, C.c_int32), _p(e, C.c_int32)))
        return b, e

    def _result(self, res: Result) -> dict:
        nb, nf, na = res.n_blocks, res.n_frags, res.n_aln
        blocks = np.ctypeslib.as_array(C.cast(res.blocks, C.POINTER(C.c_uint8)), shape=(nb * BLOCK_DT.itemsize,)).view(BLOCK_DT).copy() if nb else np.zeros(0, BLOCK_DT)
        recs = np.ctypeslib.as_array(C.cast(res.recs, C.POINTER(C.c_uint8)), shape=(nf * REC_DT.itemsize,)).view(REC_DT).copy() if nf else np.zeros(0, REC_DT)
        frags = expand_recs(recs)
        a1 = np.ctypeslib.as_array(C.cast(res.aln1, C.POINTER(C.c_uint8)), shape=(na,)).copy() if na else np.zeros(0, np.uint8)
        a2 = np.ctypeslib.as_array(C.cast(res.aln2, C.POINTER(C.c_uint8)), shape=(na,)).copy() if na else np.zeros(0, np.uint8)
        return dict(blocks=blocks, frags=frags, aln1=a1, aln2=a2)

    def raw_result(self) -> Result:
        """The gsa_result view (pointers into library-owned memory, no copies)."""
        res = Result()
        self._ck(self.lib.gsa_get_blocks(self.ctx, C.byref(res)))
        return res

    def block_records(self) -> np.ndarray:
        """uint8 [n_blocks, 40]: the finished gsa_block records (copied; small)."""
        res = self.raw_result()
        if not res.n_blocks:
            return np.zeros((0, 40), np.uint8)
        return np.ctypeslib.as_array(C.cast(res.blocks, C.POINTER(C.c_uint8)), shape=(res.n_blocks * 40,)).reshape(-1, 40).copy()

    def blocks(self) -> dict:
        res = Result()
        self._ck(self.lib.gsa_get_blocks(self.ctx, C.byref(res)))
        return self._result(res)

    def blocks_as_dump(self, with_aln: bool = False) -> dict:
        """Same keys/shapes as oracle_py._StageReader.blocks(): records gathered in block order."""
        return result_as_dump(self.blocks(), with_aln)

    def dump_stages(self, upto: int = 8) -> dict:
        d = {}
        for st in range(1, upto + 1):
            self.run_to(st)
            if st == 1:
                q, l, r = self.seeds(); b, e = self.groups()
                d.update(s1_qpos=q, s1_qlen=l, s1_rpos=r, s1_gbeg=b, s1_gend=e)
            else:
                for k, v in self.blocks_as_dump(with_aln=(st == 8)).items():
                    d[f"s{st}_{k}"] = v
        return d

    def counters(self) -> np.ndarray:
        c = np.zeros(8, np.uint64)
        self._ck(self.lib.gsa_get_counters(self.ctx, _p(c, C.c_uint64)))
        return c

    def timings(self) -> np.ndarray:
        t = np.zeros(8, np.float32)
        self._ck(self.lib.gsa_get_timings(self.ctx, _p(t, C.c_float)))
        return t

    # ---- leaf operators ----
    def bwt_search_batch(self, start, stop):
        start = np.ascontiguousarray(start, np.int32); stop = np.ascontiguousarray(stop, np.int32)
        n = start.size
        ln = np.zeros(n, np.int32); fr = np.zeros(n, np.int32); loc = np.zeros(n * 100, np.int64)
        self._ck(self.lib.gsa_bwt_search_batch(self.ctx, C.c_int32(n), _p(start, C.c_int32), _p(stop, C.c_int32), _p(ln, C.c_int32), _p(fr, C.c_int32), _p(loc, C.c_int64)))
        return ln, fr, loc.reshape(n, 100)

    def ksw2_batch(self, s1_list, s2_list):
        """list of bytes pairs -> list of forward op strings (bytes)."""
        n = len(s1_list)
        l1 = np.array([len(x) for x in s1_list], np.int32); l2 = np.array([len(x) for x in s2_list], np.int32)
        o1 = np.concatenate([[0], np.cumsum(l1[:-1], dtype=np.int64)]).astype(np.int64) if n else np.zeros(0, np.int64)
        o2 = np.concatenate([[0], np.cumsum(l2[:-1], dtype=np.int64)]).astype(np.int64) if n else np.zeros(0, np.int64)
        mn = (l1 + l2).astype(np.int64)
        oo = np.concatenate([[0], np.cumsum(mn[:-1])]).astype(np.int64) if n else np.zeros(0, np.int64)
        p1 = b"".join(s1_list) + b"\0"; p2 = b"".join(s2_list) + b"\0"
        ops = np.zeros(int(mn.sum()) + 1, np.uint8); ol = np.zeros(n, np.int32)
        self._ck(self.lib.gsa_ksw2_batch(self.ctx, C.c_int32(n), p1, _p(o1, C.c_int64), _p(l1, C.c_int32), p2, _p(o2, C.c_int64), _p(l2, C.c_int32),
                                         ops.ctypes.data_as(C.c_char_p), _p(oo, C.c_int64), _p(ol, C.c_int32)))
        return [ops[oo[i]:oo[i] + ol[i]].tobytes() for i in range(n)]

    def gap_similarity_batch(self, q1, q2, r1, r2):
        q1 = np.ascontiguousarray(q1, np.int32); q2 = np.ascontiguousarray(q2, np.int32)
        r1 = np.ascontiguousarray(r1, np.int64); r2 = np.ascontiguousarray(r2, np.int64)
        res = np.zeros(q1.size, np.int32)
        self._ck(self.lib.gsa_gap_similarity_batch(self.ctx, C.c_int32(q1.size), _p(q1, C.c_int32), _p(q2, C.c_int32), _p(r1, C.c_int64), _p(r2, C.c_int64), _p(res, C.c_int32)))
        return res

    def align_bundle(self, contigs) -> list:
        """gsa_align_bundle: several contigs (uint8 arrays, or DeviceContig objects -- all of one kind) in ONE pass; one result dict
        per contig, each what align_contig of that contig alone gives."""
        n = len(contigs)
        on_dev = n > 0 and isinstance(contigs[0], DeviceContig)
        self._q = contigs
        qs = (C.c_char_p * n)(*[C.cast(c.ptr if on_dev else c.ctypes.data, C.c_char_p) for c in contigs])
        ql = (C.c_int32 * n)(*[int(c.size) for c in contigs])
        res = (Result * n)()
        self.lib.gsa_align_bundle.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32, C.c_uint32, C.POINTER(Result)]
        self._ck(self.lib.gsa_align_bundle(self.ctx, qs, ql, n, 2 if on_dev else 0, res))
        return [self._result(res[k]) for k in range(n)]

    def align_contig_raw(self, seq: np.ndarray) -> Result:
        """gsa_align_contig without copying the result out (views into library-owned memory)."""
        self._q = seq
        res = Result()
        self._ck(self.lib.gsa_align_contig(self.ctx, seq.ctypes.data_as(C.c_char_p), C.c_int32(seq.size), C.byref(res)))
        return res

    def close(self):
        if self.ctx:
            self.lib.gsa_destroy(self.ctx); self.ctx = C.c_void_p()
        for p in self._pinned:
            self.lib.gsa_host_free(p)
        self._pinned = []
        for d in self._devbufs:
            d.free()
        self._devbufs = []


def apply_ops(s1: bytes, s2: bytes, ops: bytes):
    """Forward M/D/I string -> the two gapped strings (what ksw2_alignment leaves in s1/s2)."""
    a, b, i, j = bytearray(), bytearray(), 0, 0
    for o in ops:
        if o == 0x44:      # 'D': gap in s1
            a.append(0x2D); b.append(s2[j]); j += 1
        elif o == 0x49:    # 'I': gap in s2
            a.append(s1[i]); b.append(0x2D); i += 1
        else:
            a.append(s1[i]); b.append(s2[j]); i += 1; j += 1
    return bytes(a), bytes(b)
