#!/usr/bin/env python3
"""bench.py -- aligned query Gbp/s of the HIP hot path (S1-S7) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = the hot path on every contig of ONE query genome: H2D of the contig from (pinned) host memory -> seed search ->
locate -> sort -> chaining -> refinement -> gap DP -> gapped strings -> block records, gap records and strings back in host memory
(D2H).  That is SURVEY 8(d)'s definition -- S1-S7 "including H2D of the query and D2H of block records" -- and it is what `value`
/ `ms_per_step` / `roofline` are measured on since round 4 (rounds 1-2 likewise; round 3 had the contigs resident in HBM).  The same
K steps with the contigs ALREADY RESIDENT in HBM (gsa_align_contig_device) are reported beside it as `resident`; the library hides the
upload of a contig behind the stages of the contig in front of it (two query slots per context, gsa_prefetch_contig), so the two
should be within a few per cent of each other.  The K steps rotate over DISTINCT query genomes (different mutation seeds: a step
never finds its own data warm in the Infinity Cache) and are handed to the library's per-contig loop (gsa_align_many) in genome order:
`--inflight` contexts per GPU (gsa_clone: one device index, one host thread per context inside the library).

Workload at N=1 (default `--workload human_full`): BASELINE.json configs[4] -- the largest configuration, and it fits one GPU
(110 GB of HBM): 24 contigs with GRCh38 chromosome lengths (3.08 Gbp) with the repeat injection of SURVEY 8(d), query = 1 %-diverged
copy (>= 98 % identity), -alen 5000.  In the same run, as `extra_workloads` (a process each): `human` (configs[3] on one GPU: one
250 Mb chromosome), `ecoli` (configs[1] stand-in, 5 Mb, 2 %), `yeast` (configs[2]: 16 contigs, 12 Mb, 2 %, -sen), `adversarial`
(250 Mb with repeat families up to 10^5 copies, microsatellites, N runs: not a BASELINE config, the regime real T2T sequence is
closer to).  Hosts below 256 GB of memory cannot build the 3.08 Gbp index: the default falls back to `human` there and says so.

N>1 (the driver launches `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`; started without a launcher
this script re-executes itself under it): configs[4] as written -- ONE genome per step, its contigs dealt to the ranks by
longest-processing-time-first (shard.assign_contigs), index replicated, no data-path collective; every rank's finished contigs
(block records, 16-byte gap/seed records, gapped strings) are gathered on rank 0 over RCCL INSIDE the timed region
(shard.ResultStage / gather_staged: device to device, exact sizes).  The job's bases are counted once: `scaling: "strong"`.
`--split`: configs[3] -- ONE contig per step, seeded by chunk range on all ranks, hits to the owner over RCCL.

Rank 0 prints ONE JSON line.  "roofline" = algorithmic bytes of the WHOLE path by the section-8(d) formula (event counters of an
accounting pass) / mean step time / 8 TB/s; "kernels" = the longest kernels with their own algorithmic bytes and live hipEvent /
stage-timer durations; "cpu_baseline" = the real reference (oracle/_ref) on this host on a bounded sample.
"""
import argparse
import ctypes as C
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
    # name: genome length(s), divergence, repeat injection, aligner parameters, distinct query genomes, contexts in flight per GPU (round 6: since the seed kernel leaves room on the CUs it
    # holds, eight contexts beat four on the unique-text workloads -- 92.4 against 94.5 ms per human genome, 39.8 against 37.8 Gbp/s on 250 Mb contigs -- and lose on the repeat-rich ones)
    "human": dict(lengths=[250_000_000], div=0.01, repeats=True, params={}, n_query=4, inflight=8, steps=160,
                  label="human-chr1-sized pair (BASELINE configs[3] on one GPU): 250 Mb reference with repeat injection (300-bp family over 10 %, 150-copy tandem) vs 1 %-diverged query, defaults"),
    "ecoli": dict(lengths=[5_000_000], div=0.02, repeats=False, params={}, n_query=4, inflight=2, steps=200,
                  label="E. coli-sized pair (BASELINE configs[1] stand-in): 5 Mb reference vs 2 %-diverged query, default -slen 15 -ind 25"),
    "yeast": dict(lengths=[1000 * k for k in YEAST_KB], div=0.02, repeats=False, params=dict(sen=1, clr=50), n_query=2, inflight=6, steps=60,
                  label="S. cerevisiae-sized pair (BASELINE configs[2]): 16 contigs / 12 Mb vs 2 %-diverged copy, -sen"),
    # BASELINE configs[4]: the whole job of the 8-GPU configuration (the index is replicated per GPU there, so one GPU
    # holds exactly this index).  6.2 G BWT rows: the >= 2^32-row device layout and the 64-bit suffix sorter on their real input.
    "human_full": dict(lengths=[1_000_000 * m for m in GRCH38_MB], div=0.01, repeats=True, params=dict(alen=5000), n_query=2, inflight=8, steps=10,
                       label="full-human-sized pair (BASELINE configs[4]): 24 contigs with GRCh38 chromosome lengths, 3.08 Gbp, repeat injection, vs 1 %-diverged copy, -alen 5000"),
    # not a BASELINE config: the repeat regime of real (T2T) sequence -- csrc/host/synth.cpp: eight families with a copy-number spectrum up
    # to 10^5 copies at 1-15 % divergence over 25 % of the sequence, microsatellites, two Mb-long N runs, soft-masked blocks
    "adversarial": dict(lengths=[250_000_000], div=0.01, repeats="adversarial", params={}, n_query=2, inflight=4, steps=32,
                        label="adversarial repeats (not a BASELINE config): 250 Mb reference, 25 % in eight repeat families up to 10^5 copies at 1-15 % divergence, microsatellites, N runs, soft-masked blocks, vs 1 %-diverged query"),
    # not a BASELINE config either: the interspersed-repeat spectrum of a primate genome over ~45 % of the sequence (Alu-like family in three age classes,
    # truncated L1-like copies, LTR-like families, ancient repeats, segmental duplications, microsatellites) -- the headline workload's realistic sibling
    "human_like": dict(lengths=[250_000_000], div=0.01, repeats="human_like", params={}, n_query=2, inflight=4, steps=32,
                       label="human-like repeats (not a BASELINE config): 250 Mb reference with a primate-like interspersed-repeat spectrum over ~45 % of the sequence (Alu-, L1-, LTR-like families by age class, ancient repeats, segmental duplications, microsatellites, N runs, soft-masked blocks), vs 1 %-diverged query"),
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


def build_reference(tmp, name, wl, rank, world, args):
    """Reference FASTA + index files (rank 0 builds, the others wait) -> (prefix, loaded index, reference contigs)."""
    from gsalign_amd import hostlib, indexio, synth
    if args.fasta_ref:                                  # real genomes supplied on the GPU host: same driver
        refs = synth.read_fasta(args.fasta_ref)
        px = os.path.join(tmp, "fasta_" + os.path.basename(args.fasta_ref)); src = args.fasta_ref
    else:
        refs = []
        for i, ln in enumerate(wl["lengths"]):
            r = synth.fast_genome(int(ln), 11000 + i)
            if wl["repeats"] == "adversarial":
                synth.inject_adversarial(r, 11000 + i)
            elif wl["repeats"] == "human_like":
                synth.inject_human_like(r, 11000 + i)
            elif wl["repeats"]:
                synth.inject_repeats(r, 11000 + i)
            refs.append((f"chr{i + 1}", r))
        px = os.path.join(tmp, f"{name}_{sum(wl['lengths'])}"); src = px + ".fa"
    done = px + ".done"
    if rank == 0:
        if not os.path.exists(done):
            if not args.fasta_ref:
                synth.write_fasta(src, refs)
            hostlib.build_index(src, px)
            open(done, "w").close()
    else:
        while not os.path.exists(done):
            time.sleep(0.1)
    return px, indexio.load_index(px), refs


def make_queries(wl, refs, args):
    """n_query distinct query genomes (lists of contigs), each a differently mutated copy of the reference -- the SAME genomes on
    every rank (the ranks share one job)."""
    from gsalign_amd import synth
    if args.fasta_query:
        return [[s for _, s in synth.read_fasta(args.fasta_query)]]
    out = []
    for k in range(wl["n_query"]):
        out.append([synth.fast_mutate(r, wl["div"], 7000 + 10 * k + i) for i, (_, r) in enumerate(refs)])
    return out


class Runner:
    """`inflight` contexts on one GPU sharing one device index; the contigs of the steps go through gsa_align_many (one host
    thread per context inside the library, contigs handed out in the order given)."""

    def __init__(self, idx, device, inflight, params):
        from gsalign_amd import capi
        self.ctx = [capi.Aligner(idx, device=device, **params)]
        for _ in range(inflight - 1):
            self.ctx.append(self.ctx[0].clone())

    def close(self):
        for c in self.ctx[1:]:
            c.close()
        self.ctx[0].close()

    def run(self, contigs, on_result=None, bundle=True, prefetch=True):
        from gsalign_amd import capi
        capi.align_many(self.ctx, contigs, on_result=on_result, in_order=True, bundle=bundle, prefetch=prefetch)


def measure(name, wl, args, tmp, rank, world, local_rank, sync, steps, warmup, dev, dist):
    """One workload: returns the per-rank measurements (dict).  A step = every contig of one query genome; with world > 1 a rank
    aligns the contigs shard.assign_contigs deals it and the results are gathered on rank 0 inside the timed region."""
    from gsalign_amd import capi, shard
    px, idx, refs = build_reference(tmp, name, wl, rank, world, args)
    genomes = make_queries(wl, refs, args)
    inflight = args.inflight if args.inflight > 0 else wl.get("inflight", 2)
    run = Runner(idx, local_rank, inflight, wl["params"])
    g0 = run.ctx[0]
    mine = [shard.assign_contigs([c.size for c in gq], world)[rank] for gq in genomes]           # contig indices of this rank, per genome
    pinned = [[g0.pinned_copy(gq[i]) for i in own] for gq, own in zip(genomes, mine)]            # this rank's contigs in pinned host memory (gsa_host_alloc), like a loader's buffers
    bp_job = float(np.mean([sum(c.size for c in gq) for gq in genomes]))                         # bases of one step of the whole JOB (all ranks)
    single_short = all(len(gq) == 1 for gq in genomes) and max(c.size for gq in genomes for c in gq) <= 16_000_000      # one short contig per step (configs[1])

    def step_list(n, src):
        out = []
        for s in range(n):
            out.extend(src[s % len(src)])
        return out

    # -- every context meets the LONGEST contig of this rank once (untimed): a context sizes its device buffers by the largest contig it has seen, and with eight contexts
    #    the warm-up steps do not show every context the long ones -- the first chr1 a context met inside the timed steps cost it a hipFree / hipMalloc of gigabytes there
    #    (measured: 112.7 ms per step against 92.4 with every context sized).  A long-running host is in this state after its first genome.
    if not single_short:
        longest = max((c for gq in pinned for c in gq), key=lambda c: c.size, default=None)
        if longest is not None and longest.size > 16_000_000:
            for g in run.ctx:
                g.align_contig_raw(longest)
    # -- accounting pass (untimed): the event counters of SURVEY 8(d), exact, averaged over the distinct query genomes (this rank's contigs)
    cnt = np.zeros(8, np.float64)
    g0.set_profiling(True, count_blocks=True)
    for gq in pinned:
        for c in gq:
            g0.align_contig_raw(c); cnt += g0.counters().astype(np.float64)
    cnt /= len(pinned)
    # -- stage split (untimed, one context alone): hipEvent stage timers; the wall time of these calls is the one-contig-at-a-time latency
    g0.set_profiling(True)
    tm = np.zeros(8, np.float64); occ_read = 0.0
    for gq in pinned:
        for c in gq:
            g0.align_contig_raw(c); tm += g0.timings().astype(np.float64); occ_read += float(g0.counters()[7])
    tm /= len(pinned); occ_read /= len(pinned)
    res = g0.raw_result(); n_blocks, n_frags, n_aln = int(res.n_blocks), int(res.n_frags), int(res.n_aln)
    for g in run.ctx:
        g.set_profiling(False)
    lat = []
    for rep in range(3):                  # one contig alone, plain gsa_align_contig, H2D inside: the latency a host sees that hands over one contig at a time
        for gq in pinned:
            for c in gq[:1]:
                t0 = time.perf_counter(); g0.align_contig_raw(c); lat.append((time.perf_counter() - t0, c.size))
    lat_ms = 1000.0 * float(np.median([t for t, _ in lat[len(lat) // 3:]])) if lat else None
    lat_bp = int(lat[-1][1]) if lat else 0

    # -- result gather (world > 1): a rank other than 0 stages every finished contig on its GPU (pinned -> device), rank 0 receives
    lib = g0.lib
    stage = shard.ResultStage(dev, upload=lambda dst, src, n: lib.gsa_device_upload(local_rank, C.c_void_p(dst), C.c_void_p(src), n)) if world > 1 else None
    per_step = [len(p) for p in pinned]
    max_items = max(len(x) for gm in [shard.assign_contigs([c.size for c in gq], world) for gq in genomes] for x in gm)
    pool = [None]; gathered = [0, 0]

    def timed_run(n, src, bundle=True, prefetch=True):
        """n steps: this rank's contigs of every step through the library's per-contig loop; world > 1: + the gather, step by step."""
        order = []                      # (step, contig index) of every entry of the flat list
        for s in range(n):
            order.extend((s, ci) for ci in mine[s % len(mine)])
        cb = None
        if stage is not None and rank != 0:
            def cb(k, r):
                s, ci = order[k]
                stage.put(s, ci, [(C.cast(r.blocks, C.c_void_p).value or 0, 40 * r.n_blocks), (C.cast(r.recs, C.c_void_p).value or 0, 16 * r.n_frags),
                                  (C.cast(r.aln1, C.c_void_p).value or 0, r.n_aln), (C.cast(r.aln2, C.c_void_p).value or 0, r.n_aln)])
                return 0
        run.run(step_list(n, src), on_result=cb, bundle=bundle, prefetch=prefetch)
        if stage is not None:
            for s in range(n):
                got, pool[0] = shard.gather_staged(stage.take(s), max_items, device=dev, host_pool=pool[0])
                gathered[0] += len(got); gathered[1] += sum(int(b.size) for b in got)

    # (priming, untimed: the timed call's own shape once -- gsa_align_many sizes its bundles of short contigs by the work it is
    #  handed, and a context that meets a larger pass than it has seen grows its device buffers: hipMalloc inside a timed step)
    bundle_main = not single_short       # configs[1] is ONE 5 Mb contig: a real run cannot bundle it with anything -- `value` = every contig in a pass of its own
    short = min(c.size for gq in genomes for c in gq) <= 16_000_000          # (contigs short enough to be bundled: the shape of the passes depends on how many are queued)
    if max(len(gq) for gq in genomes) > 1 or single_short:
        timed_run(steps if short else min(steps, 2 * len(pinned)), pinned, bundle=bundle_main)
    timed_run(warmup, pinned, bundle=bundle_main)
    # timed region: the dominant kernel (seed search) is timed live, two hipEvents per contig on the library's stream; the
    # library sums them per context (gsa_get_timings, kernel_ms[6])
    for g in run.ctx:
        g.set_profiling(False, seed_only=True)
    gathered[0] = gathered[1] = 0
    sync(); t0 = time.perf_counter()
    timed_run(steps, pinned, bundle=bundle_main)
    sync(); t_total = time.perf_counter() - t0
    seed_live_ms = sum(float(g.timings()[6]) for g in run.ctx) / max(1, steps)      # per step (= this rank's contigs of one query genome), beside the other contexts' kernels
    for g in run.ctx:
        g.set_profiling(False)
    side = {}
    if world == 1 and not args.no_side_legs:
        # the same K steps with the contigs already RESIDENT in HBM (gsa_align_contig_device) -- secondary figure
        resident = [[g0.device_copy(c, local_rank) for c in gq] for gq in pinned]
        timed_run(min(warmup, 2), resident, bundle=bundle_main)
        sync(); t0 = time.perf_counter(); timed_run(steps, resident, bundle=bundle_main); sync(); side["t_resident"] = time.perf_counter() - t0
        # ... and uploaded when their turn comes (no prefetch): what the overlap buys
        timed_run(min(warmup, 2), pinned, bundle=bundle_main, prefetch=False)
        sync(); t0 = time.perf_counter(); timed_run(steps, pinned, bundle=bundle_main, prefetch=False); sync(); side["t_noprefetch"] = time.perf_counter() - t0
        if single_short:
            # many 5 Mb genomes queued at once: gsa_align_many aligns ~12 of them per pass (bundles) -- batch throughput, NOT one E. coli run
            timed_run(steps, pinned, bundle=True)
            sync(); t0 = time.perf_counter(); timed_run(steps, pinned, bundle=True); sync(); side["t_bundled"] = time.perf_counter() - t0
    parity_gpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        # the GPU's answer for the sample the CPU baseline is about to time (first cpu_sample / 4 bases of the longest contig of genome 0 against
        # the FULL index): cpu_baseline() compares the reference's stage-8 result with it -- parity at the scale the headline is quoted on
        q0 = max(genomes[0], key=lambda c: c.size)
        n1 = int(min(q0.size, max(1000, args.cpu_sample // 4)))
        g0.align_contig(np.ascontiguousarray(q0[:n1]))
        parity_gpu = (n1, g0.blocks_as_dump(with_aln=True))
    run.close()
    alg = {"occ_blocks": 64.0 * cnt[0], "lf_steps": 64.0 * cnt[1], "sa_reads": 8.0 * cnt[2], "query": float(np.mean([sum(c.size for c in gq) for gq in pinned])), "seeds": 16.0 * cnt[3],
           "dp_cells": cnt[4], "dp_fragments": cnt[6]}
    if world > 1:           # the whole job's counters: sum over the ranks' shards
        import torch
        v = torch.tensor(list(alg.values()) + list(cnt), dtype=torch.float64, device=dev); dist.all_reduce(v, op=dist.ReduceOp.SUM)
        v = v.cpu().numpy(); alg = dict(zip(alg.keys(), (float(x) for x in v[:len(alg)]))); cnt = v[len(alg):]
    return dict(px=px, refs=refs, genomes=genomes, t_total=t_total, bp=bp_job * steps, bp_per_step=bp_job, steps=steps, alg=alg, cnt=cnt, tm=tm,
                occ_read=occ_read, seed_live_ms=seed_live_ms, side=side, n_blocks=n_blocks, n_frags=n_frags, n_aln=n_aln, contigs_per_step=len(genomes[0]),
                contigs_this_rank=per_step, inflight=inflight, lat_ms=lat_ms, lat_bp=lat_bp, single_short=single_short, short=short, gathered=gathered, bundle_main=bundle_main,
                parity_gpu=parity_gpu)


def measure_split(name, wl, args, tmp, rank, world, local_rank, sync, steps, warmup, dev):
    """--split (BASELINE configs[3]: ONE chr1-sized contig on N GPUs): every rank seeds its chunk range of the same contig
    (gsa_seed_chunks), the hits go to the owner (shard.exchange_hits: point-to-point over RCCL), the owner chains, extends
    and holds the result (gsa_finish_contig); the owner rotates with the step.  Strong scaling: the job's bases are counted once."""
    from gsalign_amd import shard
    px, idx, refs = build_reference(tmp, name, wl, rank, world, args)
    genomes = make_queries(wl, refs, args)                     # the SAME query contigs on every rank
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
    return dict(t_total=t_total, bp=bp_per_step * steps, bp_per_step=bp_per_step, steps=steps)


PMC_FILE = "profiles/r06_pmc_{name}.json"


def pmc_traffic(name):
    """PMC traffic per step of the top kernels.  NOT measured in this run: rocprofv3 --pmc needs passes of its own (one counter
    set per pass, tools/pmc_top.sh -> tools/pmc_top.py -> profiles/r06_pmc_<workload>.json, which names the commit it was taken
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
    # ONE timer kind for every entry: the hipEvent stage timers of ONE context ALONE (untimed pass in front of the timed region), so an entry is
    # the time its kernels need with the chip to themselves and entries add up to (about) a contig's latency -- never to more than the step.
    # The seed kernels are also timed LIVE in the timed region (two events per contig, summed over the contexts in flight): those durations
    # overlap each other and the other contexts' kernels, i.e. they are occupancy, not work -- reported as `live_sum_ms_per_step` beside the
    # entry, with no fraction derived from it.
    chain_alg = 2.0 * m["alg"]["seeds"]          # chain + refine read the located seeds and write the refined ones: 16 B each way per seed
    for kname, ms, ab, key, live in (("k_seed_wg + dense kernels (seed search, S1)" + (" [rank 0's shard]" if world > 1 else ""), float(tm[0]), seed_alg / world, "k_seed_wg", m["seed_live_ms"] if m["seed_live_ms"] > 0 else None),
                                     ("k_seed_select + sort + group (locate/order, S1 tail; algorithmic bytes = dense-SA reads + seed records, the reference's LF walk is replaced, not performed)" + (" [rank 0's shard]" if world > 1 else ""), float(tm[1] + tm[2]), loc_alg / world, "k_seed_select", None),
                                     ("fused look-back passes + window walk + gap similarity (chain + refine, S2-S5; algorithmic bytes = 16 B per seed read + 16 B written)" + (" [rank 0's shard]" if world > 1 else ""), float(tm[3] + tm[4]), chain_alg / world, "chain_refine", None),
                                     ("k_dp_stripe + k_dp_small/lane + k_materialize (extend stage, S6-S7)" + (" [rank 0's shard]" if world > 1 else ""), float(tm[5]), dp_alg / world, "k_dp_stripe", None)):
        a = ab / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        tr = None
        if pmc and key in pmc.get("kernels", {}):
            tr = float(pmc["kernels"][key]["traffic_bytes_per_step"])
        ent = {"kernel": kname, "timer": "hipEvent stage timer, one context alone, untimed pass", "ms_per_step": ms, "algorithmic_bytes_per_step": ab, "achieved": a, "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "traffic": tr,
               "physical_frac": (tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (tr is not None and ms > 0) else None}
        if live is not None:
            ent["live_sum_ms_per_step"] = live
            ent["live_sum_note"] = "sum over the contexts in flight of this kernel's hipEvent durations inside the TIMED steps: overlapping launches (occupancy), not comparable with ms_per_step of the step"
        if a > HBM_PEAK_GBS:
            # more "algorithmic" bytes per second than HBM can move: the kernel does not read them (small reference: nearly every
            # search is settled by the k-mer table and one text comparison instead of the Occ walk the formula counts, see
            # counters_per_step.occ_blocks_read) -- so this is no roofline of the kernel and no fraction is claimed for it
            ent["frac"] = None
            ent["note"] = "achieved > peak: the formula's bytes (the reference's Occ walk) are not read by this kernel; no fraction claimed"
        kern.append(ent)
    traffic = float(pmc["traffic_bytes_per_step"]) if pmc and "traffic_bytes_per_step" in pmc else None
    traffic_source = (f"{PMC_FILE.format(name=name)} (separate rocprofv3 --pmc passes at commit {pmc.get('commit', '?')}, --inflight 1; not measured in this run)" if pmc else None)

    def leg(t, note):
        ms = 1000.0 * t / m["steps"]
        return {"value": m["bp_per_step"] / (ms * 1e-3) / 1e9 if ms > 0 else None, "ms_per_step": ms, "unit": "Gbp/s", "note": note}
    out = {
        "value": total_bp / t_max / 1e9, "ms_per_step": ms_step,
        "config": {"workload": wl["label"] + f"; {len(m['genomes'])} distinct query genomes rotating; step = every contig of one genome: H2D of the contig from pinned host memory (hidden behind the previous contig: gsa_prefetch_contig) + S1..S7 + D2H of records and strings"
                   + ("" if m["bundle_main"] else "; every contig in a pass of its own (GSA_MANY_NO_BUNDLE: one genome = one short contig, nothing to bundle it with)")
                   + ("; the K genomes are queued at once and gsa_align_many lets short contigs of consecutive genomes share passes (bundles of <= 64 Mb)" if m["bundle_main"] and m["short"] else ""),
                   "query_bp_per_step": int(m["bp_per_step"]), "contigs_per_step": m["contigs_per_step"], "inflight_contexts_per_gpu": m["inflight"],
                   "parallelism": (f"contig-shard x{world} (LPT, shard.assign_contigs), index replicated, results gathered on rank 0 over RCCL inside the timed region" if world > 1 else "one GPU"),
                   "aligner_params": wl["params"]},
        "roofline": {"bound": "hbm", "kernel": "whole hot path S1-S7 incl. query H2D (SURVEY 8(d) formula: 64 N_occblk + 64 N_lf + 8 N_sa + L_query + 16 N_seed + N_dpcells + sum(m+n))",
                     "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBS * world), "traffic": traffic, "traffic_source": traffic_source,
                     "physical_frac": (traffic / (ms_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)) if traffic is not None else None,
                     "physical_frac_note": "PMC traffic per step / measured step time / peak: what HBM actually moves; `frac` divides the SURVEY 8(d) formula's bytes (the reference's Occ and LF walks, most of which this path does not perform) by the same time",
                     "algorithmic_bytes_per_step": alg_total, "terms": m["alg"], "bytes_per_query_base": alg_total / m["bp_per_step"],
                     # the DOMINANT KERNEL by itself (k_seed_wg + the dense kernels of a contig = one "launch"): its algorithmic bytes per launch / its average duration, measured
                     # LIVE in the timed region (two hipEvents per contig on the library's stream, beside the other contexts' kernels -- so this is occupancy time, the lower bound of its rate)
                     "dominant_kernel": "k_seed_wg (+ dense kernels), seed search S1", "launches_per_step": m["contigs_per_step"] if world == 1 else None,
                     "avg_launch_ms": (m["seed_live_ms"] / m["contigs_per_step"]) if (m["seed_live_ms"] > 0 and world == 1) else None,
                     "dominant_kernel_frac": (seed_alg / (m["seed_live_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if (m["seed_live_ms"] > 0 and world == 1) else None},
        "kernels": kern,
        "stage_ms_one_context_alone": {"seed_search": float(tm[0]), "locate": float(tm[1]), "sort_group": float(tm[2]), "chain": float(tm[3]), "refine": float(tm[4]),
                                       "extend": float(tm[5]), "host_lists": float(tm[7])},
        "one_contig_latency": {"ms": m["lat_ms"], "contig_bp": m["lat_bp"], "note": "gsa_align_contig of ONE contig alone (first contig of the genome), H2D inside, nothing else on the GPU: median wall time"},
        "counters_per_step": {"occ_blocks_algorithmic": int(cnt[0]), "occ_blocks_read": int(m["occ_read"]), "lf_steps_algorithmic": int(cnt[1]), "hits": int(cnt[2]), "seeds": int(cnt[3]),
                              "dp_cells": int(cnt[4]), "dp_jobs": int(cnt[5]), "blocks_last_contig": m["n_blocks"], "records_last_contig": m["n_frags"], "string_bytes_last_contig": m["n_aln"]},
    }
    sd = m["side"]
    if "t_resident" in sd:
        out["resident"] = leg(sd["t_resident"], "same K steps with the contigs ALREADY RESIDENT in HBM (gsa_align_contig_device): no H2D of the query")
        out["h2d_inclusive_over_resident"] = out["value"] / out["resident"]["value"] if out["resident"]["value"] else None
    if "t_noprefetch" in sd:
        out["no_prefetch"] = leg(sd["t_noprefetch"], "same K steps, every contig uploaded when its turn comes (GSA_MANY_NO_PREFETCH): what hiding the upload buys")
    if "t_bundled" in sd:
        out["bundled"] = leg(sd["t_bundled"], "BATCH throughput, not one genome: the K one-contig genomes queued at once, gsa_align_many aligns ~12 contigs per pass (bundles of <= 64 Mb)")
    if world > 1:
        out["gather"] = {"contigs_received_by_rank0": m["gathered"][0], "bytes_received_by_rank0": m["gathered"][1]}
    return out


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


def parity_compare(want, parity_gpu, n1):
    """The CPU side's stage-8 result of the sample against the GPU's (blocks, records, both gapped-string pools): (verdict, detail)."""
    if parity_gpu is None:
        return "not compared", "no GPU result of the sample"
    if parity_gpu[0] != n1:
        return "not compared", f"sample lengths differ ({parity_gpu[0]} vs {n1})"
    got = parity_gpu[1]
    for k in sorted(want):
        if k not in got or got[k].shape != want[k].shape or not np.array_equal(got[k], want[k]):
            return "DIFFERENT", f"{k}: GPU {got[k].shape if k in got else None} vs CPU {want[k].shape}"
    return "identical", f"{int(want['b_score'].size)} blocks, {int(want['f_qpos'].size)} records, {int(want['aln1'].size)} string bytes per side"


def cpu_baseline(px, qry, tmp, budget_bp, params=None, parity_gpu=None):
    """The real reference (oracle/_ref) on this host, on a bounded sample of the same workload (the first budget_bp bases of
    one query contig against the FULL index): (a) its hot path S1-S7 at one thread through libgsref -- whose stage-8 result is KEPT and
    compared with the GPU's result for the same bases (`parity_sample`); (b) the unmodified CLI with all cores, whole program, and the
    same with a 1 kb query -- the difference is its hot path + output at N threads."""
    from gsalign_amd import synth
    from oracle import oracle_py as op
    sample = qry[:budget_bp]
    params = dict(params or {})
    qfa = os.path.join(tmp, "cpu_q.fa"); tiny = os.path.join(tmp, "cpu_tiny.fa")
    synth.write_fasta(qfa, [("q", sample)]); synth.write_fasta(tiny, [("t", sample[:1000])])
    cores = os.cpu_count() or 1
    n1 = int(min(sample.size, max(1000, budget_bp // 4)))     # (one thread gets a quarter of the sample: ~5 s of hot path)
    if not op.have_ref():
        from gsalign_amd import indexio
        o = op.Oracle(indexio.load_index(px), params)
        o.set_query(sample[:n1]); t = time.time(); o.run_to(8); dt = time.time() - t
        verdict, detail = parity_compare(o.blocks(with_aln=True), parity_gpu, n1); o.close()
        return {"value": n1 / dt / 1e9, "unit": "Gbp/s", "cores": 1, "kind": "port", "sample": f"first {n1} bp of one query contig, oracle restatement S1-S7, 1 thread, {dt:.2f} s",
                "parity_sample": verdict, "parity_detail": detail + f"; oracle restatement vs gsa_align_contig, first {n1} bp against the full index"}
    dump = os.path.join(tmp, "cpu_parity.npz")
    code = ("import sys,time;sys.path.insert(0,%r);import numpy as np;from oracle import oracle_py as op;from gsalign_amd import synth;"
            "r=op.RefLib(%r,%r);q=synth.read_fasta(%r)[0][1][:%d];r.set_query(q);t=time.time();r.run_to(8);print(time.time()-t);np.savez(%r,**r.blocks(with_aln=True))" % (ROOT, px, params, qfa, n1, dump))
    t1 = float(subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    with np.load(dump) as z:
        verdict, detail = parity_compare({k: z[k] for k in z.files}, parity_gpu, n1)
    os.remove(dump)
    nthr = physical_cores()                        # the reference's pthread path on all physical cores of this host
    t = time.time(); op.ref_run_cli(px, qfa, os.path.join(tmp, "cpu_out"), threads=nthr); tn = time.time() - t
    t = time.time(); op.ref_run_cli(px, tiny, os.path.join(tmp, "cpu_out0"), threads=nthr); t0 = time.time() - t
    v1 = n1 / t1 / 1e9
    thot = max(tn - t0, 1e-3); vn = sample.size / thot / 1e9
    best_cores, best = (1, v1) if v1 >= vn else (nthr, vn)
    return {"value": best, "unit": "Gbp/s", "cores": best_cores, "kind": "reference", "parity_sample": verdict,
            "parity_detail": detail + f"; the real reference's stage-8 result (libgsref, -t 1) vs gsa_align_contig, first {n1} bp of the longest contig of query genome 0 against the full index, parameters {params}",
            "sample": f"one query contig vs the full index; reference hot path S1-S7 at -t 1 (libgsref) on its first {n1} bp: {t1:.2f} s = {v1:.5f} Gbp/s; "
                      f"unmodified reference CLI -t {nthr} on its first {sample.size} bp: {tn:.2f} s whole program, {t0:.2f} s with a 1 kb query (index load + unpack) -> {thot:.2f} s for hot path + output = {vn:.5f} Gbp/s; "
                      f"host has {cores} logical cores"}


def end_to_end(px, genome, params, inflight, tmp, tag):
    """The PRODUCT end to end on this workload ("End-to-end wall time is reported beside it"; the reference prints its own at
    GSAlign.cpp:550): `GSAlign_hip -i <index> -q <query genome 0 as FASTA> -o <prefix> -ctx N -timing` in a process of its own -- index files
    from disk, FASTA parse, gsa_create, the hot path, MAF and VCF written to disk -- and where its wall time went (the CLI's own clock)."""
    from gsalign_amd import hostlib, synth
    qfa = os.path.join(tmp, f"e2e_{tag}_q.fa"); outp = os.path.join(tmp, f"e2e_{tag}_out")
    synth.write_fasta(qfa, [(f"q{i + 1}", c) for i, c in enumerate(genome)])
    cmd = [hostlib.CLI_PATH, "-i", px, "-q", qfa, "-o", outp, "-ctx", str(inflight), "-timing"]
    for k, v in (params or {}).items():
        if k == "sen":
            if v:
                cmd.append("-sen")
        elif k == "one":
            if v:
                cmd.append("-one")
        elif k == "clr" and (params or {}).get("sen"):
            continue                     # (-sen sets -clr 50 itself)
        else:
            cmd += ["-" + k, str(v)]
    t = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    wall = time.time() - t
    if os.environ.get("GSA_DUMP_BUFFERS"):      # (diagnosis: the CLI lists its contexts' device buffers on stderr)
        with open(os.path.join(ROOT, "gpurun_out", f"e2e_{tag}_stderr.txt"), "w") as f:
            f.write(r.stderr)
    line = [ln for ln in r.stderr.splitlines() if ln.startswith("GSA_TIMING ")]
    out = {"command": " ".join(["GSAlign_hip"] + cmd[1:]), "rc": r.returncode, "wall_s": wall}
    if r.returncode == 0 and line:
        d = json.loads(line[-1][len("GSA_TIMING "):])
        out.update(d)
        out["note"] = ("wall time of the whole program incl. process start; index_load = .bwt/.sa/.pac from disk (page cache) + unpacking, gsa_create = index upload + device-side tables, "
                       "query_load runs beside both; align_many = the hot path with the contigs read from pageable host memory; MAF / variants are formatted while later contigs align, "
                       "output_drain_after_align = from the last contig's alignment to both output files closed (what was left of MAF formatting / writing, the VCF beside the MAF writer's tail, the contexts' teardown beside both); vcf = sort + format + write (inside output_drain); gbp_per_s_excl_index_build = query bases / total")
    else:
        out["stderr_tail"] = r.stderr[-600:]
    for ext in (".maf", ".vcf", ".aln"):
        try:
            os.remove(outp + ext)
        except OSError:
            pass
    if not os.environ.get("GSA_BENCH_KEEP"):      # (tools/e2e_threads.sh runs the CLI again on the same files)
        os.remove(qfa)
    return out


COMPACT_LIMIT = 4096         # bytes: the LAST stdout line must stay well inside what the driver keeps of stdout (round 5: a 32 KB line was cut, parsed = null)
DETAIL_FILE = os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def _clean(x):
    """NaN / Infinity are not JSON: -> null, recursively."""
    if isinstance(x, float):
        return x if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return _clean(x.item())
    return x


def _r(x, nd=4):
    return round(float(x), nd) if isinstance(x, (int, float)) and not isinstance(x, bool) and x == x and abs(x) != float("inf") else None


def compact_line(out):
    """The driver-facing object: the contract's keys + roofline + cpu_baseline + end_to_end and one number per further workload, short
    strings only.  Everything else (terms, kernels[], stage times, counters, notes, the extras' full objects) is the detail file."""
    cfg = out.get("config", {}); rf = out.get("roofline") or {}; cb = out.get("cpu_baseline"); ee = out.get("end_to_end")
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    c["value"] = _r(c["value"], 4); c["ms_per_step"] = _r(c["ms_per_step"], 4)
    c["data"] = str(c["data"])[:80]
    c["config"] = {"workload": str(cfg.get("workload", ""))[:120]}
    for k in ("query_bp_per_step", "contigs_per_step", "inflight_contexts_per_gpu", "world_size", "backend", "contigs_on_rank0_after_gather"):
        if k in cfg:
            c["config"][k] = cfg[k]
    if "parallelism" in cfg:
        c["config"]["parallelism"] = str(cfg["parallelism"])[:60]
    if "aligner_params" in cfg:
        c["config"]["aligner_params"] = cfg["aligner_params"]
    if "fallback" in cfg:
        c["config"]["fallback"] = str(cfg["fallback"])[:100]
    if rf:
        c["roofline"] = {"bound": rf.get("bound", "hbm"), "kernel": str(rf.get("kernel", ""))[:60], "achieved": _r(rf.get("achieved"), 2), "peak": _r(rf.get("peak"), 1), "unit": rf.get("unit", "GB/s"),
                         "frac": _r(rf.get("frac"), 4), "traffic": _r(rf.get("traffic"), 0), "physical_frac": _r(rf.get("physical_frac"), 4),
                         "algorithmic_bytes_per_step": _r(rf.get("algorithmic_bytes_per_step"), 0), "dominant_kernel": str(rf.get("dominant_kernel", ""))[:40], "avg_launch_ms": _r(rf.get("avg_launch_ms"), 4),
                         "launches_per_step": rf.get("launches_per_step"), "dominant_kernel_frac": _r(rf.get("dominant_kernel_frac"), 4)}
    if cb:
        c["cpu_baseline"] = {"value": _r(cb.get("value"), 6), "unit": cb.get("unit", "Gbp/s"), "cores": cb.get("cores"), "kind": cb.get("kind"), "sample": str(cb.get("sample", ""))[:160],
                             "parity_sample": cb.get("parity_sample")}
    if ee:
        c["end_to_end"] = {k: _r(ee.get(k), 3) for k in ("total_s", "align_many_s", "gsa_create_s", "index_load_s") if k in ee}
        if "error" in ee or "stderr_tail" in ee:
            c["end_to_end"]["error"] = str(ee.get("error", ee.get("stderr_tail")))[:80]
    if "resident" in out and out["resident"]:
        c["resident_value"] = _r(out["resident"].get("value"), 4)
    st = out.get("stage_ms_one_context_alone")
    if st:
        c["stage_ms_alone"] = {k: _r(v, 2) for k, v in st.items()}
    xs = []
    for e in out.get("extra_workloads", []):
        x = {"workload": e.get("workload"), "value": _r(e.get("value"), 3), "ms_per_step": _r(e.get("ms_per_step"), 3)}
        if e.get("roofline"):
            x["frac"] = _r(e["roofline"].get("frac"), 4)
        if e.get("error"):
            x["error"] = str(e["error"])[:60]
        xs.append(x)
    if xs:
        c["extra_workloads"] = xs
    c["detail"] = "gpurun_out/bench_detail.json"
    c = _clean(c)
    line = json.dumps(c, allow_nan=False)
    for drop in ("stage_ms_alone", "extra_workloads", "resident_value", "end_to_end"):     # (never needed at today's sizes: a guard, so that the line cannot outgrow the limit again)
        if len(line) < COMPACT_LIMIT:
            break
        c.pop(drop, None); line = json.dumps(c, allow_nan=False)
    assert len(line) < COMPACT_LIMIT, len(line)
    return line


def emit(out):
    """Detail -> gpurun_out/bench_detail.json (the whole object, merged back by gpurun); compact object -> the LAST line of stdout."""
    full = _clean(out)
    try:
        os.makedirs(os.path.dirname(DETAIL_FILE), exist_ok=True)
        name = DETAIL_FILE if os.environ.get("GSA_BENCH_DETAIL") is None else os.environ["GSA_BENCH_DETAIL"]
        with open(name, "w") as f:
            json.dump(full, f)
    except OSError as e:
        print(f"bench.py: could not write the detail file: {e}", file=sys.stderr)
    sys.stdout.flush()
    print(compact_line(full), flush=True)


def dry_main(args):
    """--dry: the launcher, the rank plumbing, the LPT contig shard and the staged result gather of the multi-GPU path on CPU (gloo) with a
    stub in place of the aligner -- what tests/test_bench_launcher.py runs at world size 2.  Prints the same JSON shape; no performance claim."""
    import torch
    import torch.distributed as dist
    from gsalign_amd import capi, shard
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr); sys.exit(2)
    if args.steps <= 0:
        args.steps = 3
    if args.warmup < 0:
        args.warmup = 1
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100)                       # ONE genome, the same on every rank
    contigs = [rng.integers(65, 69, size=n).astype(np.uint8) for n in (40000, 30000, 20000, 10000, 5000)]
    mine = shard.assign_contigs([c.size for c in contigs], world)[rank]
    max_items = max(len(x) for x in shard.assign_contigs([c.size for c in contigs], world))
    dev = torch.device("cpu")
    stage = shard.ResultStage(dev)

    def stub(ci, c):
        B = np.zeros(1 + ci % 3, capi.BLOCK_DT); B["score"] = int(c.sum() % 100000); B["n_frag"] = 1
        R = np.zeros(2 + ci, capi.REC_DT)
        return B, R, c[:8 + ci].copy(), c[8:16 + 2 * ci].copy()

    def sync():
        if world > 1:
            dist.barrier()

    def one_step(s, check=False):
        got_all = []
        for ci in mine:
            B, R, a1, a2 = stub(ci, contigs[ci])
            if rank != 0 and world > 1:
                stage.put(s, ci, [(B.ctypes.data, B.nbytes), (R.ctypes.data, R.nbytes), (a1.ctypes.data, a1.nbytes), (a2.ctypes.data, a2.nbytes)])
        if world > 1:
            got, _ = shard.gather_staged(stage.take(s), max_items, device=dev)
            got_all = [shard.parse_staged(b, capi.BLOCK_DT, capi.REC_DT) for b in got]
            if check and rank == 0:
                for ci, r in got_all:
                    B, R, a1, a2 = stub(ci, contigs[ci])
                    assert np.array_equal(r["blocks"], B) and np.array_equal(r["recs"], R) and np.array_equal(r["aln1"], a1) and np.array_equal(r["aln2"], a2), ci
        return len(mine) + len(got_all)
    for s in range(args.warmup):
        one_step(s)
    sync(); t0 = time.perf_counter()
    seen = 0
    for s in range(args.steps):
        seen = one_step(s, check=True)
    sync(); t = time.perf_counter() - t0
    tt = torch.tensor([t], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(compact_line({"metric": "aligned query Gbp/s (whole node)", "value": float(sum(c.size for c in contigs) * args.steps) / float(tt.item()) / 1e9, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1000.0 * float(tt.item()) / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
                          "data": "dry run: stub aligner on CPU, gloo (launcher / plumbing check only)", "config": {"workload": "dry", "contigs_on_rank0_after_gather": seen, "world_size": world}}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = the workload's own (human_full: 10 genomes = 240 contigs)")
    ap.add_argument("--warmup", type=int, default=-1, help="-1 = steps / 8, at least 2")
    ap.add_argument("--workload", default="human_full", choices=sorted(WORKLOADS))
    ap.add_argument("--genome", type=int, default=0, help="override the reference length of a one-contig workload")
    ap.add_argument("--divergence", type=float, default=-1.0)
    ap.add_argument("--inflight", type=int, default=0, help="contexts (host threads) per GPU working on different contigs (0 = the workload's own)")
    ap.add_argument("--extra", default="human,ecoli,yeast,adversarial,human_like", help="further workloads measured in the same run (a process each, loops of their own); '' = none")
    ap.add_argument("--hwq", type=int, default=16, help="GPU_MAX_HW_QUEUES for this process (0 = leave the runtime's default of 4; the contexts in flight have 5 streams each)")
    ap.add_argument("--split", action="store_true", help="N > 1 only: ONE contig per step, its seed search sharded by chunk range over the ranks (BASELINE configs[3])")
    ap.add_argument("--no-torch", action="store_true", help="experiment (N = 1): keep torch out of the process, so that the library runs on the system's HIP runtime")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the N > 1 run (nccl = RCCL; gloo: plumbing checks)")
    ap.add_argument("--same-gpu", action="store_true", help="plumbing check on a one-GPU box: every rank uses GPU 0 (with --backend gloo)")
    ap.add_argument("--dry", action="store_true", help="no GPU: stub aligner + gloo, checks the launcher and the rank plumbing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the `end_to_end` leg (the GSAlign_hip program on query genome 0 of this workload: FASTA + index from disk -> MAF + VCF)")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the `resident` / `no_prefetch` / `bundled` legs (profiling runs)")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="bases of one query contig the CPU baseline is timed on")
    ap.add_argument("--fasta-ref", default="", help="real genomes on this host: reference FASTA (the index is built next to the bench's scratch files) ...")
    ap.add_argument("--fasta-query", default="", help="... and query FASTA; every sequence is a contig of ONE query genome")
    args = ap.parse_args()
    relaunch_if_needed(args)
    if args.dry:
        return dry_main(args)
    if args.hwq > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hwq)      # (the runtime's default is 4 hardware queues per process; `inflight` contexts x 5 streams share them)

    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr); sys.exit(2)
    if bool(args.fasta_ref) != bool(args.fasta_query):
        print("bench.py: --fasta-ref and --fasta-query go together", file=sys.stderr); sys.exit(2)
    no_torch = args.no_torch and world == 1
    if no_torch:
        # experiment: N = 1 without torch in the process -- libgsa_hip.so then runs on the system's HIP runtime (its D2H copies go through the
        # SDMA engines; under the runtime bundled with torch they are shader blits, profiles/archive/r04_blit_probe.txt).  Every library call the
        # timed region makes is synchronous, so the clock needs no device-wide synchronisation of its own
        torch = dist = dev = None
    else:
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            print("bench.py needs a GPU: libgsa_hip.so has no CPU path", file=sys.stderr); sys.exit(2)
        if args.same_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        torch.zeros(1, device=dev)                                  # (the runtime's helper threads exist after the first use of the device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world); dev = torch.device("cpu")

    tmp = os.environ.get("GSA_BENCH_TMP") or os.path.join(tempfile.gettempdir(), f"gsa_bench_{os.environ.get('MASTER_PORT', 'single')}")
    os.makedirs(tmp, exist_ok=True)
    if rank == 0:
        for fn in os.listdir(tmp):
            if fn.endswith(".done") and not os.environ.get("GSA_BENCH_KEEP"):
                os.remove(os.path.join(tmp, fn))
    if world > 1:
        dist.barrier()

    def sync():
        if torch is not None:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def whole_job(m):
        if torch is None:
            return float(m["t_total"]), float(m["bp"])
        tt = torch.tensor([m["t_total"]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()), float(m["bp"])             # (one job shared by the ranks: its bases are counted once)

    name = args.workload
    fallback = None
    if name == "human_full" and not args.fasta_ref and host_mem_gb() < 256:
        fallback = f"host has {host_mem_gb():.0f} GB of memory, the 3.08 Gbp index build needs ~120: fell back to --workload human (BASELINE configs[3] on one GPU)"
        name = "human"
    wl = dict(WORKLOADS[name])
    if args.steps <= 0:
        args.steps = wl["steps"]
    if args.warmup < 0:
        args.warmup = max(2, args.steps // 8)
    if args.fasta_ref:
        wl["label"] = f"real genomes: {os.path.basename(args.fasta_ref)} vs {os.path.basename(args.fasta_query)}"; wl["n_query"] = 1
    if args.genome > 0 and len(wl["lengths"]) == 1:
        wl["lengths"] = [args.genome]; wl["label"] += f" [reference length overridden: {args.genome}]"
    if args.divergence >= 0:
        wl["div"] = args.divergence; wl["label"] += f" [divergence overridden: {args.divergence}]"
    if args.split and world > 1 and len(wl["lengths"]) == 1:
        m = measure_split(name, wl, args, tmp, rank, world, local_rank, sync, args.steps, args.warmup, dev)
        t_max, total_bp = whole_job(m)
        if rank == 0:
            emit({"metric": "aligned query Gbp/s (whole node)", "value": total_bp / t_max / 1e9, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": 1000.0 * t_max / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                              "config": {"workload": wl["label"] + "; ONE contig per step, S1 sharded by chunk range over the ranks, hits to the (rotating) owner over RCCL, S2-S7 on the owner",
                                         "query_bp_per_step": int(m["bp_per_step"]), "parallelism": f"chunk-range shard x{world} of one contig, index replicated", "world_size": world}})
        dist.barrier(); dist.destroy_process_group()
        return
    m = measure(name, wl, args, tmp, rank, world, local_rank, sync, args.steps, args.warmup, dev, dist)
    t_max, total_bp = whole_job(m)
    out = None
    if rank == 0:
        out = {"metric": "aligned query Gbp/s (whole node)", "value": None, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
               "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic" if not args.fasta_ref else "fasta files"}
        out.update(summarise(name, wl, m, t_max, total_bp, world, args))
        out["config"]["vcf_concordance"] = "bit-identical MAF/VCF vs reference on tests/golden (tests/test_gpu_cli.py)"
        out["config"]["world_size"] = world
        if world > 1:
            out["config"]["backend"] = args.backend + (" (RCCL)" if args.backend == "nccl" else "")
        if fallback:
            out["config"]["fallback"] = fallback
    if rank == 0:
        # the further workloads (loops of their own), each in a process of its own behind the main measurement: in one
        # process the second workload runs 10-25 % slower whatever the order (measured both ways); N = 1 only
        extras = []
        for xn in [x for x in args.extra.split(",") if x and x != name and world == 1 and not args.fasta_ref]:
            st = WORKLOADS[xn]["steps"]
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", xn, "--extra", "", "--no-cpu-baseline", "--hwq", str(args.hwq)] + ([] if xn == "human" and not args.no_e2e else ["--no-e2e"])
            if args.no_torch:
                cmd.append("--no-torch")
            try:
                dfile = os.path.join(tmp, f"detail_{xn}.json")
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, GSA_BENCH_TMP=tmp, GSA_BENCH_KEEP="1", GSA_BENCH_DETAIL=dfile))
                json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])            # (the child's own compact line must parse too)
                with open(dfile) as f:
                    d = json.load(f)
                e = {"workload": xn, "steps": st, "unit": "Gbp/s"}
                e.update({k: d[k] for k in ("value", "ms_per_step", "resident", "h2d_inclusive_over_resident", "no_prefetch", "bundled", "one_contig_latency", "config", "roofline", "kernels", "stage_ms_one_context_alone", "counters_per_step", "end_to_end") if k in d})
                extras.append(e)
            except Exception as ex:      # noqa: BLE001
                extras.append({"workload": xn, "value": None, "error": repr(ex)[:300]})
        out["extra_workloads"] = extras
        if world == 1 and not args.no_e2e and not args.fasta_ref and name in ("human_full", "human", "yeast", "ecoli"):
            try:
                out["end_to_end"] = end_to_end(m["px"], m["genomes"][0], wl["params"], m["inflight"], tmp, name)
            except Exception as e:      # noqa: BLE001
                out["end_to_end"] = {"error": repr(e)[:300]}
        if not args.no_cpu_baseline and world == 1:
            try:
                q0 = max(m["genomes"][0], key=lambda c: c.size)
                out["cpu_baseline"] = cpu_baseline(m["px"], q0, tmp, args.cpu_sample, params=wl["params"], parity_gpu=m["parity_gpu"])
            except Exception as e:   # never lose the GPU line to a baseline hiccup      # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
        emit(out)
        if out.get("cpu_baseline", {}).get("parity_sample") == "DIFFERENT":      # a fast result that differs from the reference's is not a result
            print("bench.py: the GPU's result for the CPU baseline's sample differs from the reference's: " + out["cpu_baseline"]["parity_detail"], file=sys.stderr)
            if world > 1:
                dist.barrier(); dist.destroy_process_group()
            sys.exit(3)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
