"""End-to-end check on the GPU box: a 12 Mb three-contig pair through our CLI and the unmodified reference CLI (oracle/_ref),
MAF and VCF compared byte for byte.  python tools/big_cli_check.py"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
from gsalign_amd import synth
from oracle import oracle_py as op
op.build(ref=False)
assert op.have_ref(), "no oracle/_ref"
root = os.environ.get("GRAFT_REPO_ROOT", ".")
d = tempfile.mkdtemp()
refs, qrys = synth.make_pair(12000000, 3, 0.02, seed=123)
qrys[1] = (qrys[1][0], synth.revcomp(qrys[1][1]))
synth.write_fasta(os.path.join(d, "r.fa"), refs); synth.write_fasta(os.path.join(d, "q.fa"), qrys)
t0 = time.time(); subprocess.run([os.path.join(root, "gsalign_amd", "bin", "GSAlign_hip"), "-r", "r.fa", "-q", "q.fa", "-o", "mine", "-t", "1"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); t1 = time.time()
subprocess.run([op.REF_GSALIGN, "-r", "r.fa", "-q", "q.fa", "-o", "theirs", "-t", "16"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); t2 = time.time()
for ext in ("maf", "vcf"):
    a = open(os.path.join(d, "mine." + ext), "rb").read(); b = open(os.path.join(d, "theirs." + ext), "rb").read()
    print(ext, len(a), len(b), "IDENTICAL" if a == b else "DIFFERENT")
print("ours %.1f s (incl. index build), reference %.1f s" % (t1 - t0, t2 - t1))
