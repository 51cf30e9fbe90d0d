#!/usr/bin/env python3
"""Markdown tables of README.md's "Numbers" section from a full bench record (bench.py --full-out) and the committed PMC traffic
(profiles/r0N/search_traffic*.json): python scripts/readme_numbers.py profiles/r06/bench_default_box4_run1.json"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1]))
traffic = {}
for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*", "search_traffic*.json"))):
    t = json.load(open(p))
    for e in (t if isinstance(t, list) else [t]):
        w = e["workload"]
        if "moved_over_algorithmic" not in e:
            continue
        traffic[(w["nb"], w["dim"], w.get("data"), w.get("rank"), w["graph"], w["L"])] = (e, os.path.relpath(p, ROOT))
cfg = d["config"]
nb, dim = [int(x) for x in cfg["workload"].split("base ")[1].split(" ")[0].split("x")]
rf = d["roofline"]
key = rf["traffic_key"]
print("| L_pq | recall@10 | QPS | row bytes, % of 8 TB/s | moved / algorithmic bytes (rocprofv3 FETCH + WRITE) | form of the visited set |")
print("|---|---|---|---|---|---|")
forms = {"lset": "exact set in LDS", "exact_hbm": "look-ahead byte tags + hub bitmap", "filter_log": "LDS filter + id log + K4", "filter_only": "LDS filter"}
for p in d["L_pq_sweep"]:
    t = traffic.get((key["nb"], key["dim"], key["data"], key["rank"], key["graph"], p["L_pq"]))
    f = [forms.get(k, k) for k in (p.get("forms") or {}) if k != "hub_bits_log2"]
    hb = (p.get("forms") or {}).get("hub_bits_log2")
    print("| %d | %.4f | %s | %.1f | %s | %s%s |" % (p["L_pq"], p["recall_at_10"], "{:,.0f}".format(p["qps"]), p["pct_of_8000"],
                                                   ("%.3f × (%s)" % (t[0]["moved_over_algorithmic"], t[1].split("/")[1])) if t else "", ", ".join(f), (" (2^%d bits)" % hb) if hb else ""))
print()
w = d.get("roofline_worstcase") or {}
print("headline: L_pq %d, recall %.4f, %s QPS, %.3f ms/step, frac %.3f, distinct_rows_frac %.3f, frac_cache_served %.3f, frac_hbm_only %s (worst-case traffic %s)" % (
    cfg["L_pq"], cfg["recall_at_10"], "{:,.0f}".format(d["value"]), d["ms_per_step"], rf["frac"], rf["distinct_rows_frac"], rf["frac_cache_served"],
    w.get("frac"), w.get("traffic")))
cb, c1, c0 = d.get("cpu_baseline") or {}, d.get("cpu_baseline_1_thread") or {}, d.get("cpu_baseline_config1") or {}
print("cpu: 16 threads %s QPS (x%.0f), 1 thread %s; config1 cpu %s QPS gpu %s" % (cb.get("value"), cb.get("gpu_over_cpu") or 0, c1.get("value"), c0.get("value"), c0.get("gpu_qps_same_inputs")))
g = d.get("gt_build") or {}
print("gt: streamed %.4g dist/s = %.3f of peak; resident 65,536: %.3f; 10,000: %s; d512: %s" % (g.get("value", 0), g["roofline"]["frac"], (g.get("k2_device_resident") or {}).get("frac_of_mfma_peak", 0),
      (g.get("k2_small_batch") or {}).get("frac_of_mfma_peak"), g.get("k2_d512")))
print("two streams %s, host form %s" % ((d.get("two_streams_pipelined") or {}).get("qps"), (d.get("host_form_pcie_inclusive") or {}).get("qps")))
for c in d.get("configs") or []:
    r = c["roofline"]
    print("side %s: nb %d dim %d L %d recall %.4f QPS %s frac %.3f cache_served %s hbm_only %s traffic %s cpu %s | sweep %s | secs %s" % (
        c["name"], c["nb"], c["dim"], c["L_pq"], c["recall_at_k"], "{:,.0f}".format(c["value"]), r["frac"], r.get("frac_cache_served"), r.get("frac_hbm_only"), r.get("traffic"),
        (c.get("cpu_baseline") or {}).get("value"), [(p["L_pq"], round(p["pct_of_8000"], 1), round(p["recall_at_k"], 3)) for p in c["L_pq_sweep"]], {k: round(v, 1) for k, v in c["seconds"].items()}))
print("device_memory", d.get("device_memory"))
print("setup", cfg["setup_seconds"])
