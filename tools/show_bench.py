#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line (stage split, counters, roofline terms, kernels)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(e, name):
    print("==", name, "value %.3f Gbp/s  %.3f ms/step  roofline frac %.3f" % (e["value"], e["ms_per_step"], e["roofline"]["frac"]), " resident", (e.get("resident") or {}).get("value"), " no-prefetch", (e.get("no_prefetch") or {}).get("value"), " bundled", (e.get("bundled") or {}).get("value"), " latency ms", (e.get("one_contig_latency") or {}).get("ms"))
    print(" stage", {k: round(v, 3) for k, v in e["stage_ms_one_context_alone"].items()})
    print(" cnt", e["counters_per_step"])
    print(" terms MB", {k: round(v / 1e6, 1) for k, v in e["roofline"]["terms"].items()}, "B/base", round(e["roofline"]["bytes_per_query_base"], 1))
    for k in e["kernels"]:
        print("  ", k["kernel"][:28], round(k["ms_per_step"], 3), "ms", round(k["achieved"], 1), "GB/s", "traffic", k["traffic"])
show(d, "main"); 
for e in d.get("extra_workloads", []):
    if e.get("value") is None: print("==", e["workload"], e.get("error"))
    else: show(e, e["workload"])
if "cpu_baseline" in d: print(d["cpu_baseline"])
for e in [d] + d.get("extra_workloads", []):
    if "end_to_end" in e: print("== end_to_end", e.get("workload", "main"), json.dumps(e["end_to_end"]))
