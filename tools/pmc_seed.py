#!/usr/bin/env python3
"""gpurun_out/pmc/*.csv (tools/pmc_seed.sh) -> profiles/archive/r01_pmc_seed.json: HBM traffic per launch of the
dominant kernel (the non-accounting k_seed_wg) and of k_dp_stripe, gfx950 corrections applied as
MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 128-byte read requests at 64 B: doubled when the
request mix confirms it)."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, pat):
    vals = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if pat(k):
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {c: (sum(v) / len(v), len(v)) for c, v in vals.items()}


def main():
    d = os.path.join(ROOT, "gpurun_out", "pmc")
    seed = lambda k: "k_seed_wg" in k and "Lb1ELb0" not in k and "<true" not in k
    dp = lambda k: "k_dp_stripe" in k
    out = {"_what": "HBM traffic of the dominant kernel from rocprofv3 PMC passes (separate runs, --pmc only with --kernel-trace), workload = bench.py default (5 Mb, 2 %), per launch, mean over the timed launches",
           "workload": {"genome": 5000000, "divergence": 0.02}, "kernel": "k_seed_wg<false,true>"}
    res = {}
    for name, pat in (("seed", seed), ("dp_stripe", dp)):
        r = {}
        for f in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum_TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum_TCC_EA0_WRREQ_64B_sum"):
            p = os.path.join(d, f + ".csv")
            if os.path.exists(p):
                r.update({c: v[0] for c, v in per_kernel(p, pat).items()})
        res[name] = r
    s = res["seed"]
    # FETCH_SIZE is in KB and equals RDREQ x 64 B; requests that are not 32-byte ones are 128-byte on this part
    rd = s.get("TCC_EA0_RDREQ_sum", 0.0); rd32 = s.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    read_bytes = (rd - rd32) * 128.0 + rd32 * 32.0 if rd else s.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
    write_bytes = s.get("WRITE_SIZE", 0.0) * 1024.0
    out["_method"] = ("MI355X_MICROARCH.md HBM section: FETCH_SIZE = TCC_EA0_RDREQ x 64 B while the requests are 128 B wide, so reads are doubled; "
                      f"here from the request counters directly: TCC_EA0_RDREQ_sum = {rd:.0f} of which 32-byte {rd32:.0f} per launch, FETCH_SIZE = {s.get('FETCH_SIZE', 0):.0f} KB. "
                      f"WRITE_SIZE = {s.get('WRITE_SIZE', 0):.0f} KB taken as is (uncalibrated).")
    out.update({"read_bytes": int(read_bytes), "write_bytes": int(write_bytes), "traffic_bytes": int(read_bytes + write_bytes), "raw": res})
    json.dump(out, open(os.path.join(ROOT, "profiles", "r01_pmc_seed.json"), "w"), indent=1)
    print(json.dumps(out)[:1500])


if __name__ == "__main__":
    main()
