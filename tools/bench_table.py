#!/usr/bin/env python3
"""CPU: the numbers table of DESIGN.md section 6 / README.md from a bench.py detail file (gpurun_out/bench_detail.json).
usage: tools/bench_table.py <detail.json>"""
import json
import sys

d = json.load(open(sys.argv[1]))


def row(name, e):
    rf = e.get("roofline") or {}
    st = e.get("stage_ms_one_context_alone") or {}
    res = (e.get("resident") or {}).get("value")
    lat = (e.get("one_contig_latency") or {}).get("ms")
    stages = " / ".join(f"{st.get(k, 0):.2f}" for k in ("seed_search", "locate", "chain", "refine", "extend"))
    pf = rf.get("physical_frac")
    return (f"| {name} | {e['value']:.2f} | {e['ms_per_step']:.2f} | {res:.2f} |" if res else f"| {name} | {e['value']:.2f} | {e['ms_per_step']:.2f} | -- |") + \
        f" {rf.get('frac', 0) or 0:.2f}" + (f" ({pf:.2f})" if pf else "") + f" | {stages} | {lat:.2f} |" if lat else ""


print("| workload | Gbp/s (upload inside the step) | ms per step | Gbp/s, contigs resident | roofline frac (physical) | stages alone: seed / locate / chain / refine / extend (ms per step) | one contig alone (ms) |")
print("|---|---|---|---|---|---|---|")
print(row("**" + d["config"]["workload"].split(":")[0].split(" (")[0] + "** (default)", d))
for e in d.get("extra_workloads", []):
    if e.get("value") is not None:
        print(row(e["workload"], e))
cb = d.get("cpu_baseline")
if cb:
    print()
    print(f"cpu_baseline ({cb.get('kind')}): {cb.get('value'):.5f} Gbp/s on {cb.get('cores')} cores; parity of the sample: {cb.get('parity_sample')}; {cb.get('sample')}")
for e in [d] + d.get("extra_workloads", []):
    ee = e.get("end_to_end")
    if ee and "total_s" in ee:
        keys = ("total_s", "index_load_s", "gsa_create_s", "query_load_s", "align_many_s", "output_drain_after_align_s", "maf_write_s", "vcf_s", "reserve_s", "reserve_wait_s", "ref_unpack_s", "align_starts_at_s")
        print()
        print(f"end_to_end ({e.get('workload', 'default')}): " + ", ".join(f"{k} {ee[k]}" for k in keys if k in ee))
