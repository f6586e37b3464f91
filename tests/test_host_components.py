"""CPU tests of the host-side components (rows (f) of SURVEY.md section 8):
index builder -> byte-identical files; MAF/VCF emitters -> byte-identical text,
fed with the ORACLE's finished blocks (the GPU is not needed for this)."""
import ctypes as C
import filecmp
import os

import numpy as np
import pytest

from gsalign_amd import hostlib


@pytest.fixture(scope="module", autouse=True)
def built():
    hostlib.build()


@pytest.mark.parametrize("name", ["cx", "small"])
def test_index_builder_byte_identical(golden_dir, tmp_path, name):
    px = str(tmp_path / name)
    hostlib.build_index(os.path.join(golden_dir, f"{name}.ref.fa"), px)
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        assert filecmp.cmp(f"{px}.{ext}", os.path.join(golden_dir, f"{name}.{ext}"), shallow=False), f"{name}.{ext} differs from the reference's"


@pytest.mark.parametrize("name", ["cx", "small"])
def test_index_builder_64bit_suffix_sorter(golden_dir, tmp_path, name, monkeypatch):
    """References above 1 Gbp (2G + 1 >= 2^31 suffixes: BASELINE configs[4], 3.1 Gbp) take the 64-bit instance of the suffix
    sorter; GSA_INDEX_64BIT=1 forces it on the fixtures.  Same bytes as the reference's bwt_index (bwtindex.c:77-149)."""
    monkeypatch.setenv("GSA_INDEX_64BIT", "1")
    px = str(tmp_path / name)
    hostlib.build_index(os.path.join(golden_dir, f"{name}.ref.fa"), px)
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        assert filecmp.cmp(f"{px}.{ext}", os.path.join(golden_dir, f"{name}.{ext}"), shallow=False), f"{name}.{ext} differs from the reference's"


@pytest.mark.parametrize("name,wide", [("cx", False), ("small", False), ("cx", True)])
def test_index_builder_parallel_suffix_sorter(golden_dir, tmp_path, name, wide, monkeypatch):
    """Round 3: references above 1 M suffixes are sorted on all host cores (bucket by 8 bases, 64-bit keys, packed-text compares:
    index_io.cpp, parallel_suffix_sort) so that a 3.1 Gbp reference builds in minutes; GSA_INDEX_PAR_MIN sends the fixtures
    through it (32- and 64-bit instance), four threads.  Same bytes as the reference's bwt_index (bwtindex.c:77-149)."""
    monkeypatch.setenv("GSA_INDEX_PAR_MIN", "1000"); monkeypatch.setenv("GSA_INDEX_THREADS", "4")
    if wide:
        monkeypatch.setenv("GSA_INDEX_64BIT", "1")
    px = str(tmp_path / name)
    hostlib.build_index(os.path.join(golden_dir, f"{name}.ref.fa"), px)
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        assert filecmp.cmp(f"{px}.{ext}", os.path.join(golden_dir, f"{name}.{ext}"), shallow=False), f"{name}.{ext} differs from the reference's"


def test_index_builder_parallel_vs_reference_builder_on_repeats(oracle_built, tmp_path, monkeypatch):
    """The parallel sorter where its comparisons run deep: adversarial repeats (thousands of 1 %-divergent copies, a tandem array,
    microsatellites, N runs packed as random bases) in two contigs, 6 Mb -- against the reference's own bwt_index run here, and
    against our serial SA-IS builder."""
    from gsalign_amd import synth
    if not oracle_built.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    refs, _ = synth.make_adversarial_pair(6_000_000, 2, 0.0, seed=17, n_run=200_000)
    synth.inject_repeats(refs[1][1], 5, frac=0.05)                       # + the 150-copy tandem array
    fa = str(tmp_path / "r.fa"); synth.write_fasta(fa, refs)
    monkeypatch.setenv("GSA_INDEX_THREADS", "8")
    hostlib.build_index(fa, str(tmp_path / "par"))
    monkeypatch.setenv("GSA_INDEX_THREADS", "1")
    hostlib.build_index(fa, str(tmp_path / "ser"))
    oracle_built.ref_build_index(fa, str(tmp_path / "ref"))
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        assert filecmp.cmp(str(tmp_path / f"par.{ext}"), str(tmp_path / f"ref.{ext}"), shallow=False), ext
        assert filecmp.cmp(str(tmp_path / f"ser.{ext}"), str(tmp_path / f"ref.{ext}"), shallow=False), ext


@pytest.mark.parametrize("name,params,maf,vcf", [
    ("cx", {}, "cx.maf", "cx.vcf"), ("cx", dict(sen=1, clr=50), "cx_sen.maf", "cx_sen.vcf"), ("small", {}, "small.maf", "small.vcf"),
    # flag variants of the reference CLI (tests/golden/make_golden.py --cli-variants): -unique, -fmt 2, -one, -idy 95, a combination, -sen -fmt 2
    ("cx", dict(unique=1), "cx_unique.maf", "cx_unique.vcf"), ("cx", dict(fmt=2), "cx_fmt2.aln", "cx_fmt2.vcf"), ("cx", dict(one=1), "cx_one.maf", "cx_one.vcf"),
    ("cx", dict(idy=95), "cx_idy95.maf", "cx_idy95.vcf"), ("cx", dict(one=1, ind=40, clr=300, alen=1000, unique=1), "cx_combo.maf", "cx_combo.vcf"),
    ("cx", dict(sen=1, clr=50, fmt=2), "cx_sen_fmt2.aln", None)])
@pytest.mark.parametrize("par_min", [None, "1"])
def test_emitters_byte_identical(oracle_built, golden_dir, tmp_path, name, params, maf, vcf, par_min, monkeypatch):
    """par_min = "1" (GSA_HOST_PAR_MIN): every block's text lines, every block's variants, the VCF lines and the FASTA / index loaders go
    through their PARALLEL forms (host pool, csrc/host/par.h) although the goldens are small -- same bytes either way."""
    from gsalign_amd import indexio
    if par_min:
        monkeypatch.setenv("GSA_HOST_PAR_MIN", par_min)
    px = os.path.join(golden_dir, name)
    params = dict(params); unique = params.pop("unique", 0); fmt = params.pop("fmt", 1)
    o = oracle_built.Oracle(indexio.load_index(px), params)

    def per_contig(ci, seq):
        o.set_query(seq); o.run_to(8)
        return o.blocks(with_aln=True)

    out_maf, out_vcf = str(tmp_path / "o.maf"), str(tmp_path / "o.vcf")
    hostlib.emit(px, os.path.join(golden_dir, f"{name}.qry.fa"), out_maf, out_vcf, name, per_contig, allow_dup=not unique, fmt=fmt)
    o.close()
    assert open(out_maf, "rb").read() == open(os.path.join(golden_dir, maf), "rb").read()
    if vcf:
        assert open(out_vcf, "rb").read() == open(os.path.join(golden_dir, vcf), "rb").read()


def test_dotplot_golden(oracle_built, golden_dir, tmp_path, monkeypatch):
    """-dp (DotPloting.cpp:10-71) against the reference's own bytes: tests/golden/cx_dp.json.gz holds the gnuplot scripts and the data
    files the unmodified reference CLI handed to (a stub) gnuplot on the cx pair (make_golden.py --dotplot).  The emitter is fed the
    oracle's blocks per contig; scripts in contig order and every data file must be byte-identical."""
    import json
    from gsalign_amd import indexio, synth
    gold = json.load(open(os.path.join(golden_dir, "cx_dp.json")))
    px = os.path.join(golden_dir, "cx")
    o = oracle_built.Oracle(indexio.load_index(px))

    def per_contig(ci, seq):
        o.set_query(seq); o.run_to(8); return o.blocks(with_aln=True)

    monkeypatch.chdir(tmp_path)                       # the script names its files relative to the output prefix: `dpo`, as in the golden run
    scripts, data = [], {}
    for ci in range(len(synth.read_fasta(os.path.join(golden_dir, "cx.qry.fa")))):
        for fn in os.listdir("."):
            os.remove(fn)
        if hostlib.dotplot(px, os.path.join(golden_dir, "cx.qry.fa"), ci, "dpo.gp", "dpo", per_contig):
            scripts.append(open("dpo.gp").read())
            data.update({fn: open(fn).read() for fn in os.listdir(".") if fn != "dpo.gp"})
    o.close()
    assert scripts == gold["scripts"]
    assert data == gold["data"]



def test_index_builder_gives_up_on_deep_repeats(oracle_built, tmp_path, monkeypatch):
    """Round 4 (ADVICE): a reference with long EXACT duplicates -- here a 400 kb contig twice, and a 150 kb piece of it a third time -- makes the parallel
    sorter's packed-text comparisons run as deep as the copies are long; beyond its depth limit (GSA_INDEX_DEPTH, 4 M bases by default; 20 000 here) it
    gives up and the whole text goes through SA-IS: same files as the reference's own bwt_index either way."""
    from gsalign_amd import synth
    if not oracle_built.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    a = synth.fast_genome(400_000, 31)
    refs = [("c1", a), ("c2", a.copy()), ("c3", synth.fast_genome(100_000, 32)), ("c4", a[120_000:270_000].copy())]
    fa = str(tmp_path / "d.fa"); synth.write_fasta(fa, refs)
    oracle_built.ref_build_index(fa, str(tmp_path / "ref"))
    monkeypatch.setenv("GSA_INDEX_PAR_MIN", "1000"); monkeypatch.setenv("GSA_INDEX_THREADS", "4")
    for tag, depth in (("deep", None), ("fallback", "20000")):
        if depth:
            monkeypatch.setenv("GSA_INDEX_DEPTH", depth)
        hostlib.build_index(fa, str(tmp_path / tag))
        for ext in ("pac", "ann", "amb", "bwt", "sa"):
            assert filecmp.cmp(str(tmp_path / f"{tag}.{ext}"), str(tmp_path / f"ref.{ext}"), shallow=False), (tag, ext)


def _reference_loader(path):
    """LoadQueryFile / TrimChromosomeName / CheckQuerySeq (reference src/main.cpp:35-114), restated line by line."""
    out = []
    data = open(path, "rb").read()
    lines = data.split(b"\n")
    if data.endswith(b"\n"):
        lines = lines[:-1]
    for ln in lines:
        if ln == b"":
            continue
        if ln[:1] == b">":
            name = bytearray(ln[1:]); i = 0
            while i < len(name):
                if name[i:i + 1] == b"|":
                    name[i:i + 1] = b"-"
                elif name[i:i + 1] in (b" ", b"#", b":", b"=", b"\t"):
                    break
                i += 1
            out.append([bytes(name[:i]), bytearray()])
        else:
            if ln.endswith(b"\r"):
                ln = ln[:-1]
            if not all((65 <= c <= 90) or (97 <= c <= 122) for c in ln):
                return None
            out[-1][1] += ln
    return [(n, bytes(q)) for n, q in out]


@pytest.mark.parametrize("par_min", [None, "1", "37"])
def test_query_loader_matches_the_reference_loop(tmp_path, par_min, monkeypatch):
    """gsah_load_query reads the file in slices, cuts it into segments at line starts and fills the sequences from all of them at once (round 5);
    the reference reads line by line.  Awkward files -- CRLF, empty lines, no final newline, headers with cut characters, a '>' inside a header,
    a header directly behind a header, one-base lines, lines of every length around the segment cuts -- through both."""
    import ctypes as C
    if par_min:
        monkeypatch.setenv("GSA_HOST_PAR_MIN", par_min)
    lib = hostlib.load()
    lib.gsah_c_query_name.restype = C.c_char_p; lib.gsah_c_query_len.restype = C.c_longlong; lib.gsah_c_query_seq.restype = C.POINTER(C.c_char)
    rng = np.random.default_rng(5)

    def seq(n):
        return bytes(rng.choice(np.frombuffer(b"ACGTacgtNnRY", np.uint8), size=n))
    files = {
        "plain": b">c1\n" + seq(70) + b"\n" + seq(70) + b"\n" + seq(13) + b"\n>c2 description here\n" + seq(5) + b"\n",
        "crlf": b">c1 x\r\n" + seq(60) + b"\r\n" + seq(9) + b"\r\n>c|2#y\r\n" + seq(11) + b"\r\n",
        "no_final_newline": b">a\n" + seq(30) + b"\n>b\n" + seq(7),
        "empty_lines_and_empty_contigs": b">a\n\n" + seq(10) + b"\n\n\n>empty\n>b=3\n" + seq(3) + b"\n\n",
        "header_with_gt": b">a > b\n" + seq(40) + b"\n>x:y\tz\n" + seq(1) + b"\n" + seq(1) + b"\n",
        "many": b"".join(b">s%d|v\n" % i + b"".join(seq(int(rng.integers(1, 90))) + (b"\r\n" if i % 3 == 0 else b"\n") for _ in range(int(rng.integers(0, 40)))) for i in range(60)),
        "lone_cr_line": b">a\n" + seq(10) + b"\n\r\n" + seq(4) + b"\n",
    }
    bad = {"digit": b">a\nACGT\nAC1T\n", "space": b">a\nAC GT\n", "cr_inside": b">a\nAC\rGT\n", "gt_in_sequence": b">a\nACGT\nAC>GT\n"}
    for name, data in files.items():
        fn = str(tmp_path / (name + ".fa")); open(fn, "wb").write(data)
        want = _reference_loader(fn)
        err = C.create_string_buffer(256)
        n = lib.gsah_c_load_query(fn.encode(), err)
        assert n == len(want), (name, n, err.value)
        for i, (nm, sq) in enumerate(want):
            assert lib.gsah_c_query_name(i) == nm, (name, i)
            ln = lib.gsah_c_query_len(i)
            assert ln == len(sq) and C.string_at(lib.gsah_c_query_seq(i), ln) == sq, (name, i)
    for name, data in bad.items():
        fn = str(tmp_path / (name + ".fa")); open(fn, "wb").write(data)
        assert _reference_loader(fn) is None
        err = C.create_string_buffer(256)
        assert lib.gsah_c_load_query(fn.encode(), err) == -1 and b"non-alphabet" in err.value, name


def test_exact_sort_is_std_sort():
    """csrc/host/exact_sort.h (the VCF sort on many threads) against std::sort itself, element for element: random keys full of ties at
    several sizes and grains (so that the level-by-level partition phase, the per-range phase and the <= 16-element leaves all run), sorted /
    reversed / constant / organ-pipe / few-runs inputs, and the median-of-three killer that drives introsort into its heapsort fallback."""
    lib = hostlib.load()
    import ctypes as C
    lib.gsah_c_exact_sort_check.argtypes = [C.c_longlong, C.c_int, C.c_uint, C.c_int, C.c_longlong]
    for n in (0, 1, 2, 15, 16, 17, 33, 1000, 65537, 300000, 2000003):
        for distinct, grain in ((3, 64), (50, 1000), (100000, 1 << 16), (7, 5000)):
            assert lib.gsah_c_exact_sort_check(n, distinct, n + distinct, 0, grain) == 0, (n, distinct, grain)
    for pattern in (1, 2, 3, 4, 5, 6):
        for n in (17, 1000, 100000, 1500000):
            assert lib.gsah_c_exact_sort_check(n, 10, 1, pattern, 2000) == 0, (pattern, n)


def test_host_pool_back_to_back_runs():
    """HostPool::run with short jobs back to back on more threads than cores (late wake-ups are routine then): every index of every
    run executes exactly once (csrc/host/par.h: a run's state is its own object, taken by a worker under the pool's mutex)."""
    from gsalign_amd import hostlib
    lib = hostlib.load()
    lib.gsah_c_pool_stress.restype = C.c_int
    lib.gsah_c_pool_stress.argtypes = [C.c_int, C.c_int, C.c_uint]
    for threads, runs in ((32, 20000), (3, 20000)):
        assert lib.gsah_c_pool_stress(threads, runs, 7) == 0
