"""End-to-end on the GPU box: the GSAlign_hip CLI (host C++ + libgsa_hip.so) must
write MAF and VCF files byte-identical to the reference's on the golden inputs,
and identical to the live reference binary (oracle/_ref) on a fresh input."""
import os
import subprocess

import pytest

from gsalign_amd import hostlib, synth

pytestmark = pytest.mark.gpu


def run_cli(cwd, *args):
    subprocess.run([hostlib.CLI_PATH, *args], cwd=cwd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.mark.parametrize("name,extra,maf,vcf", [
    ("cx", [], "cx.maf", "cx.vcf"), ("cx", ["-sen"], "cx_sen.maf", "cx_sen.vcf"), ("small", [], "small.maf", "small.vcf"),
    # flag variants, goldens from the unmodified reference CLI (tests/golden/make_golden.py --cli-variants)
    ("cx", ["-unique"], "cx_unique.maf", "cx_unique.vcf"), ("cx", ["-fmt", "2"], "cx_fmt2.aln", "cx_fmt2.vcf"), ("cx", ["-one"], "cx_one.maf", "cx_one.vcf"),
    ("cx", ["-idy", "95"], "cx_idy95.maf", "cx_idy95.vcf"), ("cx", ["-one", "-ind", "40", "-clr", "300", "-alen", "1000", "-unique"], "cx_combo.maf", "cx_combo.vcf"),
    ("cx", ["-sen", "-fmt", "2"], "cx_sen_fmt2.aln", "cx_sen.vcf"), ("cx", ["-no_vcf"], "cx.maf", None),
    # the contigs spread over 1 / 4 contexts (gsa_align_many; the default is 2): same bytes whatever worked on them
    ("cx", ["-ctx", "1"], "cx.maf", "cx.vcf"), ("cx", ["-ctx", "4"], "cx.maf", "cx.vcf"), ("cx", ["-sen", "-ctx", "3", "-gpu", "0"], "cx_sen.maf", "cx_sen.vcf")])
def test_cli_golden(golden_dir, tmp_path, name, extra, maf, vcf):
    run_cli(golden_dir, "-i", name, "-q", f"{name}.qry.fa", "-o", str(tmp_path / "out"), "-t", "1", *extra)
    kind = maf.rsplit(".", 1)[1]
    assert open(tmp_path / f"out.{kind}", "rb").read() == open(os.path.join(golden_dir, maf), "rb").read()
    if vcf:
        assert open(tmp_path / "out.vcf", "rb").read() == open(os.path.join(golden_dir, vcf), "rb").read()
    else:
        assert not os.path.exists(tmp_path / "out.vcf")                   # -no_vcf (main.cpp:280)


def test_cli_one_contig_on_several_gpus(golden_dir, tmp_path):
    """BASELINE configs[3] in the C++ product: ONE query sequence and several GPUs listed (-gpu 0,0,0: three contexts with an index
    of their own each stand in for three GPUs on this one-GPU box) -> gsa_align_many seeds it by chunk range on all of them, the
    owner imports the hits device to device and finishes; MAF and VCF byte-identical to the one-GPU run and to the reference's."""
    qs = synth.read_fasta(os.path.join(golden_dir, "cx.qry.fa"))
    one = tmp_path / "one.fa"
    synth.write_fasta(str(one), [qs[0]])
    env = dict(os.environ, GSA_SPLIT_MIN="50000")               # (the contig is 120 kb: let it split)
    for tag, gpus in (("a", "0"), ("b", "0,0,0"), ("c", "0,0")):
        subprocess.run([hostlib.CLI_PATH, "-i", "cx", "-q", str(one), "-o", str(tmp_path / tag), "-t", "1", "-gpu", gpus], cwd=golden_dir, env=env, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref_maf = open(os.path.join(golden_dir, "cx.maf"), "rb").read()
    a = open(tmp_path / "a.maf", "rb").read()
    assert len(a) > 1000 and ref_maf.startswith(a)             # (the golden MAF of all contigs begins with this contig's part)
    for tag in ("b", "c"):
        assert open(tmp_path / f"{tag}.maf", "rb").read() == a and open(tmp_path / f"{tag}.vcf", "rb").read() == open(tmp_path / "a.vcf", "rb").read(), tag


def test_cli_dotplot_golden(golden_dir, tmp_path):
    """-dp: with a `gnuplot` on PATH (a stub that keeps what it is given -- none is installed here or where the golden was made) the CLI
    hands gnuplot the same scripts and data files as the unmodified reference CLI did (tests/golden/cx_dp.json.gz), removes its data
    files afterwards (DotPloting.cpp:69-70) and writes the same MAF."""
    import json
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    import make_golden
    cap = tmp_path / "cap"; cap.mkdir()
    bindir = make_golden.write_gnuplot_stub(str(tmp_path / "bin"))
    wd = tmp_path / "wd"; wd.mkdir()
    for fn in os.listdir(golden_dir):
        if fn.startswith("cx."):
            os.symlink(os.path.join(golden_dir, fn), wd / fn)
    env = dict(os.environ, PATH=bindir + os.pathsep + os.environ.get("PATH", ""), GSA_DP_CAPTURE=str(cap))
    subprocess.run([hostlib.CLI_PATH, "-i", "cx", "-q", "cx.qry.fa", "-o", "dpo", "-t", "1", "-dp"], cwd=wd, env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    gold = json.load(open(os.path.join(golden_dir, "cx_dp.json")))
    got = make_golden.collect_dotplot(str(cap))
    assert got["scripts"] == gold["scripts"] and got["data"] == gold["data"]
    assert not [f for f in os.listdir(wd) if "vs" in f]
    assert open(wd / "dpo.maf", "rb").read() == open(os.path.join(golden_dir, "cx.maf"), "rb").read()


def test_cli_builds_its_own_index_and_matches_live_reference(oracle_built, tmp_path):
    if not oracle_built.have_ref():
        pytest.skip("oracle/_ref not present")
    refs, qrys = synth.make_pair(400000, 3, 0.03, seed=77)
    qrys[2] = (qrys[2][0], synth.revcomp(qrys[2][1]))
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "r.fa"), refs); synth.write_fasta(os.path.join(d, "q.fa"), qrys)
    run_cli(d, "-r", "r.fa", "-q", "q.fa", "-o", "mine")                 # builds r.{bwt,sa,pac,ann,amb} itself
    os.makedirs(os.path.join(d, "refidx"))
    oracle_built.ref_build_index(os.path.join(d, "r.fa"), os.path.join(d, "refidx", "r"))
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert open(os.path.join(d, f"r.{ext}"), "rb").read() == open(os.path.join(d, "refidx", f"r.{ext}"), "rb").read(), ext
    subprocess.run([oracle_built.REF_GSALIGN, "-r", "r.fa", "-q", "q.fa", "-o", "theirs", "-t", "1"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert open(os.path.join(d, "mine.maf"), "rb").read() == open(os.path.join(d, "theirs.maf"), "rb").read()
    assert open(os.path.join(d, "mine.vcf"), "rb").read() == open(os.path.join(d, "theirs.vcf"), "rb").read()


def test_cli_config3_and_repeat_stress_vs_live_reference(oracle_built):
    """Whole programs side by side at BASELINE configs[2] size (16 contigs / 12 Mb / 2 % / -sen) and on the 12 Mb repeat-stress
    pair: GSAlign_hip (own index, three contexts) vs the unmodified reference CLI at -t 1 -- MAF and VCF byte-identical."""
    import sys
    if not oracle_built.have_ref():
        pytest.skip("oracle/_ref not present")
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_cli_check.py"), "c3,repeat,adversarial"], capture_output=True, text=True, timeout=900, env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))
    assert r.returncode == 0 and r.stdout.count("IDENTICAL") == 6, r.stdout + r.stderr


@pytest.mark.parametrize("name,extra,maf,vcf", [("cx", [], "cx.maf", "cx.vcf"), ("cx", ["-sen"], "cx_sen.maf", "cx_sen.vcf"), ("cx", ["-fmt", "2", "-idy", "70"], "cx_fmt2.aln", "cx_fmt2.vcf"),
                                                ("cx", ["-one", "-ind", "40", "-clr", "300", "-alen", "1000", "-unique"], "cx_combo.maf", "cx_combo.vcf"), ("small", [], "small.maf", "small.vcf")])
def test_reference_program_with_the_drop_in(oracle_built, golden_dir, tmp_path, name, extra, maf, vcf):
    """INTEGRATION.md section 2 as a compiled artefact (oracle/_ref/GSAlign_ref_hip, oracle/Makefile `ref_hip`): the REFERENCE'S OWN program
    -- its main(), argument parsing, FASTA and index loaders, MAF / ALN / VCF emitters, built from /root/reference in place -- with the
    eight pthread stages of GenomeComparison() (GSAlign.cpp:483-540) replaced by one gsa_align_contig call per query sequence
    (oracle/ref_hip_dropin.cpp).  Its output bytes == the unmodified reference's (the committed goldens)."""
    exe = os.path.join(os.path.dirname(oracle_built.REF_GSALIGN), "GSAlign_ref_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/GSAlign_ref_hip not built (no /root/reference at build time)")
    subprocess.run([exe, "-i", name, "-q", f"{name}.qry.fa", "-o", str(tmp_path / "out"), "-t", "1", *extra], cwd=golden_dir, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    kind = maf.rsplit(".", 1)[1]
    assert open(tmp_path / f"out.{kind}", "rb").read() == open(os.path.join(golden_dir, maf), "rb").read()
    assert open(tmp_path / "out.vcf", "rb").read() == open(os.path.join(golden_dir, vcf), "rb").read()
