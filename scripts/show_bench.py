#!/usr/bin/env python3
"""Short view of a bench.py JSON line (the last line starting with '{' of the given file)."""
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
r = json.loads(l[-1])
rf = r["roofline"]
print("value %.0f QPS  ms/step %.3f  frac %.4f  L %s  recall %.4f  hbm_only %s  cache_served %s" % (r["value"], r["ms_per_step"], rf["frac"], r["config"]["L_pq"],
      r["config"]["recall_at_10"] or 0, rf.get("frac_hbm_only"), rf.get("frac_cache_served")))
print("setup", r["config"]["setup_seconds"], "forms", rf.get("kernel_forms_of_the_batches_so_far"), "mem", r.get("device_memory"))
for p in r["L_pq_sweep"]:
    print("  L %5d  %5.1f %%  recall %.4f  evals %7.0f  qps %9.0f" % (p["L_pq"], p["pct_of_8000"], p["recall_at_10"] or 0, p["mean_evals"], p["qps"]))
w = r.get("roofline_worstcase")
print("worst", w and round(w["frac"], 4), "two_streams", r.get("two_streams_pipelined") and round(r["two_streams_pipelined"]["vs_one_stream"], 3),
      "host_form", r.get("host_form_pcie_inclusive") and round(r["host_form_pcie_inclusive"]["vs_device_resident"], 3))
g = r.get("gt_build")
if g: print("gt", round(g["roofline"]["frac"], 4), "k2 resident", g.get("k2_device_resident", {}).get("frac_of_mfma_peak"))
c = r.get("cpu_baseline")
if c: print("cpu16", c.get("value"), "x", c.get("gpu_over_cpu"))
for c in r.get("configs") or []:
    print(c["name"], "L", c["L_pq"], "qps %.0f" % c["value"], "recall@%d %.4f" % (c["recall_k"], c["recall_at_k"] or 0), "frac %.3f" % c["roofline"]["frac"],
          {k: round(v, 1) for k, v in c["seconds"].items()}, "cpu", (c["cpu_baseline"] or {}).get("value"), c["kernel_forms_of_the_batches"])
    print("    ", [(p["L_pq"], round(p["pct_of_8000"], 1), round(p["recall_at_k"] or 0, 3)) for p in c["L_pq_sweep"]])
