"""End-to-end checks on the GPU box: whole programs side by side -- our CLI (GSAlign_hip: own index builder, GPU hot path on
several contexts, own emitters) and the unmodified reference CLI (oracle/_ref, -t 1: its block order on score ties and its
S4/S5 race depend on the thread count) -- MAF and VCF compared byte for byte.
    python tools/big_cli_check.py [cases: plain,c3,repeat,adversarial]"""
import os, subprocess, sys, tempfile, time
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from gsalign_amd import synth
from oracle import oracle_py as op
op.build(ref=False)
assert op.have_ref(), "no oracle/_ref"
YEAST_KB = [230, 813, 317, 1532, 577, 270, 1091, 563, 440, 746, 667, 1078, 924, 784, 1091, 948]
cases = (sys.argv[1] if len(sys.argv) > 1 else "plain,c3,repeat").split(",")
ok = True
for case in cases:
    d = tempfile.mkdtemp()
    if case == "plain":      # 12 Mb, three contigs, 2 %, one reverse-complemented, defaults
        refs, qrys = synth.make_pair(12000000, 3, 0.02, seed=123); flags = []
        qrys[1] = (qrys[1][0], synth.revcomp(qrys[1][1]))
    elif case == "c3":       # BASELINE configs[2]: 16 contigs with the S. cerevisiae chromosome lengths, 2 %, -sen
        refs, qrys = synth.make_pair_fast(0, 16, 0.02, seed=124, lengths=[1000 * k for k in YEAST_KB]); flags = ["-sen"]
        qrys[5] = (qrys[5][0], synth.revcomp(qrys[5][1]))
    elif case == "adversarial":      # copy-number spectrum, microsatellites, N runs, soft-masked blocks: 12 Mb in three contigs, 2 %
        refs, qrys = synth.make_adversarial_pair(12000000, 3, 0.02, seed=126, n_run=300000); flags = []
        qrys[2] = (qrys[2][0], synth.revcomp(qrys[2][1]))
    else:                    # SURVEY 8(d) repeat-stress variant, 12 Mb in four contigs, 1 %
        refs, qrys = synth.make_pair_fast(12000000, 4, 0.01, seed=125, repeats=True); flags = []
    synth.write_fasta(os.path.join(d, "r.fa"), refs); synth.write_fasta(os.path.join(d, "q.fa"), qrys)
    t0 = time.time(); subprocess.run([os.path.join(root, "gsalign_amd", "bin", "GSAlign_hip"), "-r", "r.fa", "-q", "q.fa", "-o", "mine", "-ctx", "3", *flags], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); t1 = time.time()
    subprocess.run([op.REF_GSALIGN, "-i", "r", "-q", "q.fa", "-o", "theirs", "-t", "1", *flags], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); t2 = time.time()
    res = []
    for ext in ("maf", "vcf"):
        a = open(os.path.join(d, "mine." + ext), "rb").read(); b = open(os.path.join(d, "theirs." + ext), "rb").read()
        # (the VCF header names the reference as given on the command line: -r r.fa vs -i r)
        if ext == "vcf": a = a.replace(b"##reference=r.fa", b"##reference=r")
        res.append(f"{ext} {len(a)} bytes {'IDENTICAL' if a == b else 'DIFFERENT'}"); ok = ok and a == b
    print(f"{case}: {sum(q.size for _, q in qrys)} bp in {len(qrys)} contigs {' '.join(flags)}: {'; '.join(res)}; ours {t1 - t0:.1f} s whole program incl. index build, reference -t 1 {t2 - t1:.1f} s on our index", flush=True)
sys.exit(0 if ok else 1)
