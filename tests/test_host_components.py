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


@pytest.mark.parametrize("name,params,maf,vcf", [("cx", {}, "cx.maf", "cx.vcf"), ("cx", dict(sen=1, clr=50), "cx_sen.maf", "cx_sen.vcf"), ("small", {}, "small.maf", "small.vcf")])
def test_emitters_byte_identical(oracle_built, golden_dir, tmp_path, name, params, maf, vcf):
    from gsalign_amd import indexio
    px = os.path.join(golden_dir, name)
    o = oracle_built.Oracle(indexio.load_index(px), params)

    def per_contig(ci, seq):
        o.set_query(seq); o.run_to(8)
        return o.blocks(with_aln=True)

    out_maf, out_vcf = str(tmp_path / "o.maf"), str(tmp_path / "o.vcf")
    hostlib.emit(px, os.path.join(golden_dir, f"{name}.qry.fa"), out_maf, out_vcf, name, per_contig)
    o.close()
    assert open(out_maf, "rb").read() == open(os.path.join(golden_dir, maf), "rb").read()
    assert open(out_vcf, "rb").read() == open(os.path.join(golden_dir, vcf), "rb").read()
