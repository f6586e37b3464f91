#!/usr/bin/env python3
"""bench.py -- aligned query Gbp/s of the HIP hot path (S1-S7) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = the hot path on one query contig: seed search -> locate -> sort -> chaining -> refinement -> gap DP ->
gapped strings, block records and strings back in host memory (D2H).  `value` is measured with the query contigs ALREADY
RESIDENT IN HBM (gsa_align_contig_device: the contract's "inputs resident when the timed region starts"); the same K steps
are then run again from pinned host buffers (gsa_align_contig: H2D of the contig inside the step) and reported as
`pcie_inclusive` -- never as `value`.  The steps rotate over several DISTINCT query contigs (different mutation
seeds), so a step never finds its own data warm in the Infinity Cache, and they are driven the way a multi-contig run
drives the library: `--inflight` contexts per GPU (gsa_clone: one device index, one context per host thread), each
working on its own contig, so that the upload and the seed search of one contig overlap the DP tail of another.

Workload at N=1 (default `--workload human`): BASELINE.json's target configuration cut to one GPU -- a chr1-sized pair
(configs[3]): 250 Mb reference with the repeat-stress injection of SURVEY 8(d) (300-bp family over 10 % of the genome +
a >100-copy tandem array), query = 1 %-diverged copy (>= 98 % identity).  `--workload ecoli` (configs[1] stand-in, 5 Mb,
2 %) and `--workload yeast` (configs[2]: 16 contigs, 12 Mb, 2 %, -sen) are measured in the same run as
`extra_workloads` (short loops of their own).  N>1: one process per GPU (this script re-executes itself under
torch.distributed.run when it is started without one), index replicated, every rank aligns its own query contigs --
the path shards by contig with no data-path collective ("weak" scaling); one gather of block records over RCCL
outside the timed region.

Rank 0 prints ONE JSON line.  "roofline" = algorithmic bytes of the WHOLE path by the section-8(d) formula (event
counters of an accounting pass) / mean step time / 8 TB/s; "kernels" = the three longest kernels with their own
algorithmic bytes, live hipEvent / stage-timer durations and the PMC traffic from profiles/ when that file was taken
on the same workload; "cpu_baseline" = the real reference (oracle/_ref) on this host on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
YEAST_KB = [230, 813, 317, 1532, 577, 270, 1091, 563, 440, 746, 667, 1078, 924, 784, 1091, 948]      # S288C chromosomes I..XVI

GRCH38_MB = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57]      # chr1..22, X, Y

WORKLOADS = {
    # name: genome length(s), divergence, repeat injection, aligner parameters, distinct query genomes
    "human": dict(lengths=[250_000_000], div=0.01, repeats=True, params={}, n_query=4, inflight=4,
                  label="human-chr1-sized pair (BASELINE configs[3] on one GPU): 250 Mb reference with repeat injection (300-bp family over 10 %, 150-copy tandem) vs 1 %-diverged query, defaults"),
    "ecoli": dict(lengths=[5_000_000], div=0.02, repeats=False, params={}, n_query=4, inflight=2,
                  label="E. coli-sized pair (BASELINE configs[1] stand-in): 5 Mb reference vs 2 %-diverged query, default -slen 15 -ind 25"),
    "yeast": dict(lengths=[1000 * k for k in YEAST_KB], div=0.02, repeats=False, params=dict(sen=1, clr=50), n_query=2, inflight=3,
                  label="S. cerevisiae-sized pair (BASELINE configs[2]): 16 contigs / 12 Mb vs 2 %-diverged copy, -sen"),
    # BASELINE configs[4] on ONE GPU: the whole job of the 8-GPU configuration (the index is replicated per GPU there, so one GPU
    # holds exactly this index).  6.2 G BWT rows: the >= 2^32-row device layout and the 64-bit suffix sorter on their real input.
    "human_full": dict(lengths=[1_000_000 * m for m in GRCH38_MB], div=0.01, repeats=True, params=dict(alen=5000), n_query=2, inflight=4,
                       label="full-human-sized pair (BASELINE configs[4] on one GPU): 24 contigs with GRCh38 chromosome lengths, 3.08 Gbp, repeat injection, vs 1 %-diverged copy, -alen 5000"),
}


def relaunch_if_needed(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: become N ranks under torch.distributed.run."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def build_reference(tmp, name, wl, rank, world):
    """Reference FASTA + index files (rank 0 builds, the others wait) -> (prefix, loaded index, reference contigs)."""
    from gsalign_amd import hostlib, indexio, synth
    refs = []
    for i, ln in enumerate(wl["lengths"]):
        r = synth.fast_genome(int(ln), 11000 + i)
        if wl["repeats"]:
            synth.inject_repeats(r, 11000 + i)
        refs.append((f"chr{i + 1}", r))
    px = os.path.join(tmp, f"{name}_{sum(wl['lengths'])}")
    done = px + ".done"
    if rank == 0:
        if not os.path.exists(done):
            synth.write_fasta(px + ".fa", refs)
            hostlib.build_index(px + ".fa", px)
            open(done, "w").close()
    else:
        while not os.path.exists(done):
            time.sleep(0.1)
    return px, indexio.load_index(px), refs


def make_queries(wl, refs, rank):
    """n_query distinct query genomes (lists of contigs), each a differently mutated copy of the reference."""
    from gsalign_amd import synth
    out = []
    for k in range(wl["n_query"]):
        out.append([synth.fast_mutate(r, wl["div"], 7000 + 100 * rank + 10 * k + i) for i, (_, r) in enumerate(refs)])
    return out


def bind_process_to_device_socket(torch, dev_index):
    """Every thread of this process (the HIP runtime's helper threads included: they exist since the device was initialised)
    moves to the CPUs of the socket the GPU hangs off (sysfs local_cpulist).  Small contigs are ~60 short GPU operations with
    five host look-ins each: from the far socket a 5 Mb contig took 0.86 ms, from the near one 0.65 (taskset, two-socket host)."""
    if not os.environ.get("GSA_BIND"):      # opt-in: measured +-: 5 Mb contigs 0.86 -> 0.65-0.75 ms from the far socket, 250 Mb contigs 13.7 -> 14.4 ms
        return None
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bus = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        cpus = set()
        for part in open(f"/sys/bus/pci/devices/{bus}/local_cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
            except OSError:
                pass
        return bus
    except Exception:
        return None


class Runner:
    """`inflight` contexts on one GPU sharing one device index; steps are handed to them through a counter."""

    def __init__(self, idx, device, inflight, params):
        from gsalign_amd import capi
        self.ctx = [capi.Aligner(idx, device=device, **params)]
        for _ in range(inflight - 1):
            self.ctx.append(self.ctx[0].clone())

    def close(self):
        for c in self.ctx[1:]:
            c.close()
        self.ctx[0].close()

    def run(self, n_steps, step_of):
        """Align the contigs of n_steps steps (step_of(n) = flat list of contig buffers -- pinned host arrays or DeviceContig
        objects -- in step order) on the contexts: gsa_align_many, i.e. one host thread per context inside the library."""
        from gsalign_amd import capi
        capi.align_many(self.ctx, step_of(n_steps), in_order=True)      # (in step order: consecutive steps are DIFFERENT query genomes)


def measure(name, wl, args, tmp, rank, world, local_rank, sync, steps, warmup):
    """One workload: returns the per-rank measurements (dict).  A step = every contig of one query genome."""
    px, idx, refs = build_reference(tmp, name, wl, rank, world)
    genomes = make_queries(wl, refs, rank)
    inflight = args.inflight if args.inflight > 0 else wl.get("inflight", 2)
    run = Runner(idx, local_rank, inflight, wl["params"])
    g0 = run.ctx[0]
    pinned = [[g0.pinned_copy(c) for c in gq] for gq in genomes]       # contigs in pinned host memory (gsa_host_alloc), like a loader's buffers
    resident = [[g0.device_copy(c, local_rank) for c in gq] for gq in genomes]      # ... and the same contigs resident in HBM (gsa_device_alloc / gsa_device_upload)
    bp_per_step = float(np.mean([sum(c.size for c in gq) for gq in genomes]))

    def step_list(n, src=resident):
        out = []
        for s in range(n):
            out.extend(src[s % len(src)])
        return out

    # -- accounting pass (untimed): the event counters of SURVEY 8(d), exact, averaged over the distinct query genomes
    cnt = np.zeros(8, np.float64)
    g0.set_profiling(True, count_blocks=True)
    for gq in pinned:
        for c in gq:
            g0.align_contig_raw(c); cnt += g0.counters().astype(np.float64)
    cnt /= len(pinned)
    # -- stage split (untimed, one context alone): hipEvent stage timers
    g0.set_profiling(True)
    tm = np.zeros(8, np.float64); occ_read = 0.0
    for gq in pinned:
        for c in gq:
            g0.align_contig_raw(c); tm += g0.timings().astype(np.float64); occ_read += float(g0.counters()[7])
    tm /= len(pinned); occ_read /= len(pinned)
    res = g0.raw_result(); n_blocks, n_frags, n_aln = int(res.n_blocks), int(res.n_frags), int(res.n_aln)
    for g in run.ctx:
        g.set_profiling(False)
    # (priming, untimed: the timed call's own shape once -- gsa_align_many sizes its bundles of short contigs by the work it is
    #  handed, and a context that meets a larger pass than it has seen grows its device buffers: hipMalloc inside a timed step)
    if max(len(gq) for gq in genomes) > 1 or max(c.size for gq in genomes for c in gq) <= 16_000_000:
        run.run(steps, step_list)
    run.run(warmup, step_list)
    # timed region: the dominant kernel (seed search) is timed live, two hipEvents per contig on the library's stream; the
    # library sums them per context (gsa_get_timings, kernel_ms[6])
    for g in run.ctx:
        g.set_profiling(False, seed_only=True)
    sync(); t0 = time.perf_counter()
    run.run(steps, step_list)
    sync(); t_total = time.perf_counter() - t0
    seed_live_ms = sum(float(g.timings()[6]) for g in run.ctx) / max(1, steps)      # per step (= all contigs of one query genome), beside the other contexts' kernels
    # the same K steps from pinned HOST buffers: H2D of every contig inside the step (reported beside `value`, never as it)
    run.run(min(warmup, 2), lambda n: step_list(n, pinned))
    sync(); t0 = time.perf_counter()
    run.run(steps, lambda n: step_list(n, pinned))
    sync(); t_pcie = time.perf_counter() - t0
    recs = g0.block_records()
    run.close()
    per_step = len(pinned[0])
    alg = {"occ_blocks": 64.0 * cnt[0], "lf_steps": 64.0 * cnt[1], "sa_reads": 8.0 * cnt[2], "query": bp_per_step, "seeds": 16.0 * cnt[3],
           "dp_cells": cnt[4], "dp_fragments": cnt[6]}
    return dict(px=px, refs=refs, genomes=genomes, t_total=t_total, bp=bp_per_step * steps, bp_per_step=bp_per_step, steps=steps, alg=alg, cnt=cnt, tm=tm,
                occ_read=occ_read, seed_live_ms=seed_live_ms, t_pcie=t_pcie, recs=recs, n_blocks=n_blocks, n_frags=n_frags, n_aln=n_aln, contigs_per_step=per_step, inflight=inflight)


def measure_split(name, wl, args, tmp, rank, world, local_rank, sync, steps, warmup, dev):
    """--split (BASELINE configs[3]: ONE chr1-sized contig on N GPUs): every rank seeds its chunk range of the same contig
    (gsa_seed_chunks), the hits go to the owner (shard.exchange_hits: point-to-point over RCCL), the owner chains, extends
    and holds the result (gsa_finish_contig); the owner rotates with the step.  Strong scaling: the job's bases are counted once."""
    from gsalign_amd import shard
    px, idx, refs = build_reference(tmp, name, wl, rank, world)
    genomes = make_queries(wl, refs, 0)                     # the SAME query contigs on every rank
    run = Runner(idx, local_rank, 1, wl["params"]); g0 = run.ctx[0]
    pinned = [g0.pinned_copy(gq[0]) for gq in genomes]
    n_chunks = [(q.size + 9999) // 10000 for q in pinned]

    def step(s):
        k = s % len(pinned); owner = s % world
        b, e = shard.split_chunks(n_chunks[k], world)[rank]
        g0.seed_chunks(pinned[k], b, e)
        shard.exchange_hits(g0, owner, device=dev)
        if rank == owner:
            g0.finish_contig()
    for s in range(warmup):
        step(s)
    sync(); t0 = time.perf_counter()
    for s in range(steps):
        step(s)
    sync(); t_total = time.perf_counter() - t0
    bp_per_step = float(np.mean([q.size for q in pinned]))
    run.close()
    return dict(t_total=t_total, bp=bp_per_step * steps / world, bp_per_step=bp_per_step, steps=steps)


PMC_FILE = "profiles/r03_pmc_{name}.json"


def pmc_traffic(name):
    """PMC traffic per step of the top kernels.  NOT measured in this run: rocprofv3 --pmc needs passes of its own (one counter
    set per pass, tools/pmc_top.sh -> tools/pmc_top.py -> profiles/r03_pmc_<workload>.json, which names the commit it was taken
    at); the JSON line says where the number comes from (`traffic_source`)."""
    try:
        return json.load(open(os.path.join(ROOT, PMC_FILE.format(name=name))))
    except Exception:      # noqa: BLE001
        return None


def summarise(name, wl, m, t_max, total_bp, world, args):
    """JSON fields of one workload from rank 0's measurements + the whole-job time and bases."""
    ms_step = 1000.0 * t_max / m["steps"]
    alg_total = float(sum(m["alg"].values()))
    achieved = alg_total / (ms_step * 1e-3) / 1e9
    pmc = pmc_traffic(name)
    kern = []
    tm, cnt = m["tm"], m["cnt"]
    seed_alg = m["alg"]["occ_blocks"] + m["alg"]["query"]
    dp_alg = m["alg"]["dp_cells"] + m["alg"]["dp_fragments"]
    # locate + order: this kernel reads the DENSE suffix array (8 B per located hit) and writes the seeds; the LF walk the
    # reference's bwt_sa does for every hit (the `lf_steps` term of the whole-path formula) is work it does not do, so that
    # term is not its algorithmic traffic
    loc_alg = m["alg"]["sa_reads"] + m["alg"]["seeds"]
    for kname, ms, ab, key in (("k_seed_wg + k_dense_search (seed search, S1): mean launch duration over the TIMED steps, hipEvents, contexts in flight beside each other", m["seed_live_ms"] if m["seed_live_ms"] > 0 else float(tm[0]), seed_alg, "k_seed_wg"),
                               ("k_dp_stripe + k_dp_small/tiny + k_materialize (extend stage, S7): stage timer, one context alone, untimed pass", float(tm[5]), dp_alg, "k_dp_stripe"),
                               ("k_seed_select + sort + group (locate/order, S1 tail; algorithmic bytes = dense-SA reads + seed records, the reference's LF walk is replaced, not performed): stage timer, one context alone, untimed pass", float(tm[1] + tm[2]), loc_alg, "k_seed_select")):
        a = ab / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        tr = None
        if pmc and key in pmc.get("kernels", {}):
            tr = float(pmc["kernels"][key]["traffic_bytes_per_step"])
        ent = {"kernel": kname, "ms_per_step": ms, "algorithmic_bytes_per_step": ab, "achieved": a, "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "traffic": tr}
        if a > HBM_PEAK_GBS:
            # more "algorithmic" bytes per second than HBM can move: the kernel does not read them (small reference: nearly every
            # search is settled by the k-mer table and one text comparison instead of the Occ walk the formula counts, see
            # counters_per_step.occ_blocks_read) -- so this is no roofline of the kernel and no fraction is claimed for it
            ent["frac"] = None
            ent["note"] = "achieved > peak: the formula's bytes (the reference's Occ walk) are not read by this kernel; no fraction claimed"
        kern.append(ent)
    traffic = float(pmc["traffic_bytes_per_step"]) if pmc and "traffic_bytes_per_step" in pmc else None
    traffic_source = (f"{PMC_FILE.format(name=name)} (separate rocprofv3 --pmc passes at commit {pmc.get('commit', '?')}, --inflight 1; not measured in this run)" if pmc else None)
    ms_pcie = 1000.0 * m["t_pcie"] / m["steps"]
    return {
        "value": total_bp / t_max / 1e9, "ms_per_step": ms_step,
        "pcie_inclusive": {"value": m["bp_per_step"] * world / (ms_pcie * 1e-3) / 1e9 if ms_pcie > 0 else None, "ms_per_step": ms_pcie, "unit": "Gbp/s",
                           "note": "same K steps with every contig uploaded from pinned host memory inside the step (gsa_align_contig); rank 0's clock"},
        "config": {"workload": wl["label"] + f"; {len(m['genomes'])} distinct query genomes rotating, resident in HBM (gsa_align_contig_device); step = S1..S7 per contig incl. D2H of records + strings",
                   "query_bp_per_step": int(m["bp_per_step"]), "contigs_per_step": m["contigs_per_step"], "inflight_contexts_per_gpu": m["inflight"],
                   "parallelism": f"contig-shard x{world}, index replicated", "aligner_params": wl["params"]},
        "roofline": {"bound": "hbm", "kernel": "whole hot path S1-S7 (SURVEY 8(d) formula: 64 N_occblk + 64 N_lf + 8 N_sa + L_query + 16 N_seed + N_dpcells + sum(m+n))",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_step": alg_total, "terms": m["alg"], "bytes_per_query_base": alg_total / m["bp_per_step"]},
        "kernels": kern,
        "stage_ms_one_context_alone": {"seed_search": float(tm[0]), "locate": float(tm[1]), "sort_group": float(tm[2]), "chain": float(tm[3]), "refine": float(tm[4]),
                                       "extend": float(tm[5]), "host_lists": float(tm[7])},
        "counters_per_step": {"occ_blocks_algorithmic": int(cnt[0]), "occ_blocks_read": int(m["occ_read"]), "lf_steps_algorithmic": int(cnt[1]), "hits": int(cnt[2]), "seeds": int(cnt[3]),
                              "dp_cells": int(cnt[4]), "dp_jobs": int(cnt[5]), "blocks_last_contig": m["n_blocks"], "records_last_contig": m["n_frags"], "string_bytes_last_contig": m["n_aln"]},
    }


def host_mem_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemTotal"):
                return int(ln.split()[1]) / 1e6
    except Exception:      # noqa: BLE001
        pass
    return 0.0


def physical_cores():
    """Physical cores of this host (logical CPUs / SMT siblings per core); the reference's -t."""
    n = os.cpu_count() or 1
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        per = 0
        for part in sib.split(","):
            a, _, b = part.partition("-")
            per += int(b or a) - int(a) + 1
        return max(1, n // max(1, per))
    except Exception:      # noqa: BLE001
        return n


def cpu_baseline(px, qry, tmp, budget_bp):
    """The real reference (oracle/_ref) on this host, on a bounded sample of the same workload (the first budget_bp bases of
    one query contig against the FULL index): (a) its hot path S1-S7 at one thread through libgsref; (b) the unmodified CLI
    with all cores, whole program, and the same with a 1 kb query -- the difference is its hot path + output at N threads."""
    from gsalign_amd import synth
    from oracle import oracle_py as op
    sample = qry[:budget_bp]
    qfa = os.path.join(tmp, "cpu_q.fa"); tiny = os.path.join(tmp, "cpu_tiny.fa")
    synth.write_fasta(qfa, [("q", sample)]); synth.write_fasta(tiny, [("t", sample[:1000])])
    cores = os.cpu_count() or 1
    if not op.have_ref():
        from gsalign_amd import indexio
        o = op.Oracle(indexio.load_index(px))
        o.set_query(sample); t = time.time(); o.run_to(8); dt = time.time() - t; o.close()
        return {"value": sample.size / dt / 1e9, "unit": "Gbp/s", "cores": 1, "kind": "port", "sample": f"first {sample.size} bp of one query contig, oracle restatement S1-S7, 1 thread, {dt:.2f} s"}
    code = ("import sys,time;sys.path.insert(0,%r);import numpy as np;from oracle import oracle_py as op;from gsalign_amd import synth;"
            "r=op.RefLib(%r);q=synth.read_fasta(%r)[0][1];r.set_query(q);t=time.time();r.run_to(8);print(time.time()-t)" % (ROOT, px, qfa))
    t1 = float(subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    nthr = physical_cores()                        # the reference's pthread path on all physical cores of this host
    t = time.time(); op.ref_run_cli(px, qfa, os.path.join(tmp, "cpu_out"), threads=nthr); tn = time.time() - t
    t = time.time(); op.ref_run_cli(px, tiny, os.path.join(tmp, "cpu_out0"), threads=nthr); t0 = time.time() - t
    v1 = sample.size / t1 / 1e9
    thot = max(tn - t0, 1e-3); vn = sample.size / thot / 1e9
    best_cores, best = (1, v1) if v1 >= vn else (nthr, vn)
    return {"value": best, "unit": "Gbp/s", "cores": best_cores, "kind": "reference",
            "sample": f"first {sample.size} bp of one query contig vs the full index; reference hot path S1-S7 at -t 1 (libgsref): {t1:.2f} s = {v1:.5f} Gbp/s; "
                      f"unmodified reference CLI -t {nthr}: {tn:.2f} s whole program, {t0:.2f} s with a 1 kb query (index load + unpack) -> {thot:.2f} s for hot path + output = {vn:.5f} Gbp/s; "
                      f"host has {cores} logical cores"}


def dry_main(args):
    """--dry: the launcher, the rank plumbing and the two exchanges of the multi-GPU path on CPU (gloo) with a stub in place
    of the aligner -- what tests/test_bench_launcher.py runs at world size 2.  Prints the same JSON shape; no performance claim."""
    import torch
    import torch.distributed as dist
    from gsalign_amd import capi, shard
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr); sys.exit(2)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    contigs = [rng.integers(65, 69, size=n).astype(np.uint8) for n in (40000, 30000, 20000, 10000)]

    def stub(ci, c):
        B = np.zeros(1, capi.BLOCK_DT); B["score"] = int(c.sum() % 100000); B["n_frag"] = 1
        F = np.zeros(1, capi.FRAG_DT); F["qpos"] = ci
        return dict(blocks=B, frags=F, aln1=c[:8].copy(), aln2=c[8:16].copy())

    def sync():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        [stub(i, c) for i, c in enumerate(contigs)]
    sync(); t0 = time.perf_counter()
    for _ in range(args.steps):
        mine = {i: stub(i, c) for i, c in enumerate(contigs)}
    sync(); t = time.perf_counter() - t0
    tt = torch.tensor([t], dtype=torch.float64); tb = torch.tensor([float(sum(c.size for c in contigs) * args.steps)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX); dist.all_reduce(tb, op=dist.ReduceOp.SUM)
    allr = shard.gather_results({rank * 100 + i: r for i, r in mine.items()}, capi.BLOCK_DT, capi.FRAG_DT)
    assert len(allr) == (world * len(contigs) if rank == 0 else 0) or world == 1      # (only rank 0 receives: point-to-point sends, no padded all_gather)
    if rank == 0:
        print(json.dumps({"metric": "aligned query Gbp/s (whole node)", "value": float(tb.item()) / float(tt.item()) / 1e9, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1000.0 * float(tt.item()) / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
                          "data": "dry run: stub aligner on CPU, gloo (launcher / plumbing check only)", "config": {"workload": "dry", "gathered_contigs": len(allr)}}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="human", choices=sorted(WORKLOADS))
    ap.add_argument("--genome", type=int, default=0, help="override the reference length of a one-contig workload")
    ap.add_argument("--divergence", type=float, default=-1.0)
    ap.add_argument("--inflight", type=int, default=0, help="contexts (host threads) per GPU working on different contigs (0 = the workload's own: 2, 3 for the 5 Mb one)")
    ap.add_argument("--extra", default="ecoli,yeast,human_full", help="further workloads measured in the same run (short loops); '' = none; human_full (3.08 Gbp, ~3 min incl. its index) is skipped on hosts below 256 GB of memory")
    ap.add_argument("--hwq", type=int, default=8, help="GPU_MAX_HW_QUEUES for this process (0 = leave the runtime's default of 4; the contexts in flight have 4 streams each)")
    ap.add_argument("--split", action="store_true", help="N > 1 only: ONE contig per step, its seed search sharded by chunk range over the ranks (strong scaling)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the N > 1 run (nccl = RCCL; gloo: plumbing checks)")
    ap.add_argument("--same-gpu", action="store_true", help="plumbing check on a one-GPU box: every rank uses GPU 0 (with --backend gloo)")
    ap.add_argument("--dry", action="store_true", help="no GPU: stub aligner + gloo, checks the launcher and the rank plumbing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="bases of one query contig the CPU baseline is timed on")
    args = ap.parse_args()
    relaunch_if_needed(args)
    if args.dry:
        return dry_main(args)
    if args.hwq > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hwq)      # (the runtime's default is 4 hardware queues per process; `inflight` contexts x 4 streams share them)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr); sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: libgsa_hip.so has no CPU path", file=sys.stderr); sys.exit(2)
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.zeros(1, device=dev)                                  # (the runtime's helper threads exist after the first use of the device)
    bind_process_to_device_socket(torch, local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world); dev = torch.device("cpu")

    from gsalign_amd import shard
    tmp = os.environ.get("GSA_BENCH_TMP") or os.path.join(tempfile.gettempdir(), f"gsa_bench_{os.environ.get('MASTER_PORT', 'single')}")
    os.makedirs(tmp, exist_ok=True)
    if rank == 0:
        for fn in os.listdir(tmp):
            if fn.endswith(".done") and not os.environ.get("GSA_BENCH_KEEP"):
                os.remove(os.path.join(tmp, fn))
    if world > 1:
        dist.barrier()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def whole_job(m):
        tt = torch.tensor([m["t_total"]], dtype=torch.float64, device=dev); tb = torch.tensor([m["bp"]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX); dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        return float(tt.item()), float(tb.item())

    wl = dict(WORKLOADS[args.workload])
    if args.genome > 0 and len(wl["lengths"]) == 1:
        wl["lengths"] = [args.genome]; wl["label"] += f" [reference length overridden: {args.genome}]"
    if args.divergence >= 0:
        wl["div"] = args.divergence; wl["label"] += f" [divergence overridden: {args.divergence}]"
    if args.split and world > 1 and len(wl["lengths"]) == 1:
        m = measure_split(args.workload, wl, args, tmp, rank, world, local_rank, sync, args.steps, args.warmup, dev)
        t_max, total_bp = whole_job(m)
        if rank == 0:
            print(json.dumps({"metric": "aligned query Gbp/s (whole node)", "value": total_bp / t_max / 1e9, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": 1000.0 * t_max / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                              "config": {"workload": wl["label"] + "; ONE contig per step, S1 sharded by chunk range over the ranks, hits to the (rotating) owner over RCCL, S2-S7 on the owner",
                                         "query_bp_per_step": int(m["bp_per_step"]), "parallelism": f"chunk-range shard x{world} of one contig, index replicated"}}))
        dist.barrier(); dist.destroy_process_group()
        return
    m = measure(args.workload, wl, args, tmp, rank, world, local_rank, sync, args.steps, args.warmup)
    t_max, total_bp = whole_job(m)
    if world > 1:
        # once, outside the timed region: the block records of the last contig of every rank on every rank (what a merged
        # report would start from) -- keeps the RCCL path exercised, costs the metric nothing
        allrecs, _ = shard.gather_block_records(m["recs"], np.full(m["recs"].shape[0], rank, np.int32), device=dev)
        assert allrecs.shape[0] >= m["recs"].shape[0]
    out = None
    if rank == 0:
        out = {"metric": "aligned query Gbp/s (whole node)", "value": None, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic"}
        out.update(summarise(args.workload, wl, m, t_max, total_bp, world, args))
        out["config"]["vcf_concordance"] = "bit-identical MAF/VCF vs reference on tests/golden (tests/test_gpu_cli.py)"
    if rank == 0:
        # the further workloads (short loops of their own), each in a process of its own behind the main measurement: in one
        # process the second workload runs 10-25 % slower whatever the order (measured both ways); N = 1 only
        extras = []
        for name in [x for x in args.extra.split(",") if x and x != args.workload and world == 1]:
            if name == "human_full" and host_mem_gb() < 256:
                extras.append({"workload": name, "value": None, "error": f"skipped: host has {host_mem_gb():.0f} GB of memory, the 3.08 Gbp index build needs ~120"})
                continue
            st = 200 if name == "ecoli" else (3 if name == "human_full" else 12)
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(st), "--warmup", str(max(2, st // 5)), "--extra", "", "--no-cpu-baseline", "--hwq", str(args.hwq)]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, GSA_BENCH_TMP=tmp, GSA_BENCH_KEEP="1"))
                d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                e = {"workload": name, "steps": st, "unit": "Gbp/s"}
                e.update({k: d[k] for k in ("value", "ms_per_step", "pcie_inclusive", "config", "roofline", "kernels", "stage_ms_one_context_alone", "counters_per_step")})
                extras.append(e)
            except Exception as ex:      # noqa: BLE001
                extras.append({"workload": name, "value": None, "error": repr(ex)[:300]})
        out["extra_workloads"] = extras
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(m["px"], m["genomes"][0][0], tmp, args.cpu_sample)
            except Exception as e:   # never lose the GPU line to a baseline hiccup      # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
        print(json.dumps(out))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
