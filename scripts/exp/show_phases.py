#!/usr/bin/env python3
"""Print the k1_phases.py JSON files given on the command line as one table."""
import json
import sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print("==", f, d.get("lib"))
    for r in d["rows"]:
        ps = r.get("phase_share", {})
        extra = ""
        if "spec_hit_rate" in r:
            extra = " spec %.2f/hop hit %.3f" % (r.get("spec_tries_per_hop", 0), r["spec_hit_rate"])
        if "merge_split" in r:
            extra += " valid/hop %.1f chunks/hop %.1f merge[%s]" % (r["valid_cands_per_hop"], r["chunks_moved_per_hop"], " ".join("%s=%.3f" % kv for kv in r["merge_split"].items()))
        print("L %4d m%d qps %8.0f GBps %5.0f (%.1f%%) perf/dist %.2f cyc/hop %6.0f fresh/hop %4.1f%s  %s" % (
            r["L_pq"], r["visited"], r["qps"], r["alg_GBps"], r["alg_GBps"] / 80.0, r["evals_performed"] / max(r["distinct_evals"], 1),
            r.get("cycles_per_hop", 0), r.get("fresh_per_hop", 0), extra, " ".join("%s=%.3f" % (k, v) for k, v in ps.items())))
