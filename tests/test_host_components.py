"""CPU tests of the host-side components (rows (f) of SURVEY.md section 8):
index builder -> byte-identical files; MAF/VCF emitters -> byte-identical text,
fed with the ORACLE's finished blocks (the GPU is not needed for this)."""
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


@pytest.mark.parametrize("name,params,maf,vcf", [
    ("cx", {}, "cx.maf", "cx.vcf"), ("cx", dict(sen=1, clr=50), "cx_sen.maf", "cx_sen.vcf"), ("small", {}, "small.maf", "small.vcf"),
    # flag variants of the reference CLI (tests/golden/make_golden.py --cli-variants): -unique, -fmt 2, -one, -idy 95, a combination, -sen -fmt 2
    ("cx", dict(unique=1), "cx_unique.maf", "cx_unique.vcf"), ("cx", dict(fmt=2), "cx_fmt2.aln", "cx_fmt2.vcf"), ("cx", dict(one=1), "cx_one.maf", "cx_one.vcf"),
    ("cx", dict(idy=95), "cx_idy95.maf", "cx_idy95.vcf"), ("cx", dict(one=1, ind=40, clr=300, alen=1000, unique=1), "cx_combo.maf", "cx_combo.vcf"),
    ("cx", dict(sen=1, clr=50, fmt=2), "cx_sen_fmt2.aln", None)])
def test_emitters_byte_identical(oracle_built, golden_dir, tmp_path, name, params, maf, vcf):
    from gsalign_amd import indexio
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


def test_dotplot_script_and_data(oracle_built, golden_dir, tmp_path):
    """-dp (DotPloting.cpp:10-71).  The reference only plots when a gnuplot binary is installed (main.cpp:324, GSAlign.cpp:546)
    -- there is none here, so no golden exists; the script and the data files are checked against the blocks they are made from."""
    from gsalign_amd import indexio, synth
    px = os.path.join(golden_dir, "cx")
    idx = indexio.load_index(px)
    o = oracle_built.Oracle(idx)
    dump = {}

    def per_contig(ci, seq):
        o.set_query(seq); o.run_to(8); dump.update(o.blocks(with_aln=True)); return dump

    pre = str(tmp_path / "plot")
    assert hostlib.dotplot(px, os.path.join(golden_dir, "cx.qry.fa"), 2, pre + ".gp", pre, per_contig)      # q3_bridge: two reference sequences
    o.close()
    qname = synth.read_fasta(os.path.join(golden_dir, "cx.qry.fa"))[2][0]
    gp = open(pre + ".gp").read()
    assert gp.startswith("set terminal postscript color solid 'Courier' 15\nset output '%s-%s.ps'\n" % (pre, qname)) and "set xlabel 'Query (%s)'" % qname in gp
    plotted = [ln for ln in gp.splitlines() if ln.startswith("plot ")][0]
    score = np.zeros(len(idx.chr_names), np.int64)
    np.add.at(score, dump["b_chr"], dump["b_score"])
    want_chr = [i for i in np.argsort(-score, kind="stable") if score[i] >= 1000][:5]
    assert len(want_chr) == 2 and plotted.count(" with lp ls ") == 2
    off = np.concatenate([[0], np.cumsum(dump["b_nfrag"])])
    for rank, ci in enumerate(want_chr):
        name = idx.chr_names[ci]
        assert "'%s.%svs%s' title '%s' with lp ls %d" % (pre, qname, name, name, rank + 1) in plotted
        rows = open("%s.%svs%s" % (pre, qname, name)).read().split("\n\n")
        assert rows[0] == "0 0\n0 0"
        segs = [tuple(int(x) for x in r.split()) for r in rows[1:] if r.strip()]
        blk = [b for b in range(dump["b_score"].size) if dump["b_chr"][b] == ci and dump["b_score"][b] > 0]
        assert len(segs) == len(blk)
        for (q0, g0, q1, g1), b in zip(segs, blk):
            f0, f1 = off[b], off[b + 1] - 1
            assert q0 == dump["f_qpos"][f0] + 1 and q1 == dump["f_qpos"][f1] + dump["f_qlen"][f1] and g0 == dump["b_gpos"][b]
