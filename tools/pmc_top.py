#!/usr/bin/env python3
"""gpurun_out/pmc_<workload>/*.csv (tools/pmc_top.sh) -> profiles/r06_pmc_<workload>.json: HBM traffic per step of the
top kernels and of the whole step, from separate rocprofv3 --pmc passes.  Reads: from the request counters
(TCC_EA0_RDREQ: requests that are not 32-byte ones are 128 bytes wide on gfx950 -- the same correction as
"FETCH_SIZE x 2" in MI355X_MICROARCH.md, HBM section); writes: WRITE_SIZE (KB) as is (uncalibrated there)."""
import csv, json, os, subprocess, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = sys.argv[1] if len(sys.argv) > 1 else "human"
D = os.path.join(ROOT, "gpurun_out", f"pmc_{W}")

GROUPS = {      # key in the json -> predicate on the (mangled) kernel name
    "k_seed_wg": lambda k: ("k_seed_wg" in k and "Lb1ELb0" not in k and "<true" not in k) or "k_dense_search" in k or "k_dense_resolve" in k,
    "k_dp_stripe": lambda k: "k_dp_stripe" in k,
    "k_dp_small": lambda k: "k_dp_small" in k or "k_dp_tiny" in k or "k_dp_lane" in k,
    "k_seed_select": lambda k: "k_seed_select" in k,
    "k_materialize": lambda k: "k_materialize" in k,
    "copies": lambda k: "copyBuffer" in k or "fillBuffer" in k,
    # chain + refine (S2-S5): everything between the located hits and the leaf table -- the fused look-back passes of those stages, the sort, the
    # PosDiff-bitmap kernels, the window walk and histogram kernels, the gap-similarity kernel (the passes of stages 6-7 belong to `extend`)
    "chain_refine": lambda k: (("k_lb_pass" in k and not any(o in k for o in ("OpDpJobs", "OpSlots", "OpClassify"))) or any(n in k for n in ("k_pd_", "k_rs_", "k_walk", "k_window", "k_outlier", "k_multihit", "k_next_window", "k_gapsim", "k_leaf_emit", "k_group_keys", "k_gather_active"))),
    "extend_passes": lambda k: any(o in k for o in ("OpDpJobs", "OpSlots", "OpClassify", "k_gap_class")),
}


def commit():
    """The commit the counters belong to: the tree gpurun shipped = HEAD (+ '-dirty' when the work tree differs)."""
    if os.environ.get("GSA_PMC_COMMIT"):          # (the counters were taken on a gpurun snapshot of an earlier commit than the tree this script runs in)
        return os.environ["GSA_PMC_COMMIT"]
    try:
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True).stdout.strip()
        return h + ("-dirty" if dirty else "")
    except Exception:      # noqa: BLE001
        return "?"


def load(fn):
    per = defaultdict(lambda: defaultdict(float)); launches = defaultdict(int); seen = set()
    p = os.path.join(D, fn + ".csv")
    if not os.path.exists(p):
        return per, launches
    for r in csv.DictReader(open(p)):
        k, c = r["Kernel_Name"], r["Counter_Name"]
        per[k][c] += float(r["Counter_Value"])
        if "Launches" in r:            # (reduced on the GPU box: tools/pmc_reduce.py)
            launches[k] = max(launches[k], int(r["Launches"]))
            continue
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key); launches[k] += 1
    return per, launches


def main():
    acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(int)
    for fn in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum_TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum_TCC_EA0_WRREQ_64B_sum"):
        per, ln = load(fn)
        for k, d in per.items():
            for c, v in d.items():
                acc[k][c] += v
        for k, v in ln.items():
            launches[k] = max(launches[k], v)
    # runs of the hot path in one bench invocation = launches of k_seed_select (one per contig in every mode);
    # runs with the production seed kernel = launches of k_seed_wg<false,*>
    sel = sum(v for k, v in launches.items() if "k_seed_select" in k)
    prod = sum(v for k, v in launches.items() if "k_seed_wg" in k and "Lb1ELb0" not in k and "<true" not in k) or sum(v for k, v in launches.items() if "k_dense_resolve" in k)
    contigs_per_step = {"human": 1, "ecoli": 1, "yeast": 16, "human_full": 24, "adversarial": 1, "human_like": 1}[W]
    def bytes_of(d):
        rd, rd32 = d.get("TCC_EA0_RDREQ_sum", 0.0), d.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        rb = (rd - rd32) * 128.0 + rd32 * 32.0 if rd else d.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
        return rb, d.get("WRITE_SIZE", 0.0) * 1024.0
    out = {"_what": f"HBM traffic per step of bench.py --workload {W} from rocprofv3 --pmc passes (one counter set per pass, --kernel-trace only beside it), the workload's own --inflight (bench.py WORKLOADS: eight contexts on human / human_full since round 6)",
           "_method": "reads = (TCC_EA0_RDREQ - RDREQ_32B) x 128 B + RDREQ_32B x 32 B (gfx950: FETCH_SIZE tallies 128-byte requests at 64 B, MI355X_MICROARCH.md HBM section; CALIBRATED in round 4 -- profiles/archive/r04_pmc_calibration.txt: a random read of 16, 32, 64 or 128 bytes costs exactly one RDREQ, none of the 32-byte kind, and FETCH_SIZE counts it as 64 B: every L2 miss fetches one 128-byte line); writes = WRITE_SIZE KB x 1024 (uncalibrated); "
                      "Infinity-Cache hits are counted, so this is L2-miss traffic, an upper bound of HBM bytes; per step = total over the run / hot-path runs x contigs per step "
                      "(seed kernels: / runs with the production seed kernel)",
           "commit": commit(), "workload": W, "hot_path_runs": sel, "production_seed_runs": prod, "kernels": {}}
    tot_r = tot_w = 0.0
    for k, d in acc.items():
        rb, wb = bytes_of(d)
        if "Lb1ELb0" in k or "k_seed_wg<true" in k or "k_count_lf" in k or "k_build" in k or "k_densify" in k or "k_pack_ref" in k or "k_pres_from" in k or "k_unpack_pac" in k or "k_occ_" in k:
            continue      # accounting build / index upload: not part of a step
        tot_r += rb; tot_w += wb
    for name, pred in GROUPS.items():
        rb = wb = 0.0; n = 0
        for k, d in acc.items():
            if pred(k):
                r_, w_ = bytes_of(d); rb += r_; wb += w_; n += launches[k]
        runs = prod if name == "k_seed_wg" else sel
        if n and runs:
            out["kernels"][name] = {"launches": n, "read_bytes_per_step": rb / runs * contigs_per_step, "write_bytes_per_step": wb / runs * contigs_per_step,
                                    "traffic_bytes_per_step": (rb + wb) / runs * contigs_per_step}
    if sel:
        out["traffic_bytes_per_step"] = (tot_r + tot_w) / sel * contigs_per_step
    json.dump(out, open(os.path.join(ROOT, "profiles", f"r06_pmc_{W}.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
