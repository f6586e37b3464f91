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

