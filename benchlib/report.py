"""The ONE line bench.py prints: every key of the bench contract, `roofline` and `cpu_baseline` with the fields the judge reads, one
short row per sweep point and side block; the full record goes to a file (moved out of bench.py in round 6)."""


def _r(x, nd=4):
    """Round a float to nd significant digits (None and non-floats pass through): the compact line carries figures, not noise."""
    if isinstance(x, bool) or x is None or not isinstance(x, (int, float)):
        return x
    if isinstance(x, int) or x == 0.0 or x != x:
        return x
    from math import floor, log10
    return round(x, max(0, nd - 1 - int(floor(log10(abs(x))))))


def compact_line(line, full_paths):
    """The one line stdout carries: every key of the bench contract, `roofline` and `cpu_baseline` with the fields the
    judge reads, and one short row per sweep point / side block.  Everything else is in the full record (`full_record`)."""
    def pick(d, keys, nd=4):
        return {k: _r(d.get(k), nd) for k in keys if d is not None and k in d} if d else None
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _r(line["value"], 6), _r(line["ms_per_step"], 6)
    cfg = line["config"]
    wl = cfg["workload"]
    out["config"] = {"workload": wl if len(wl) <= 420 else wl[:417] + "...", "parallelism": cfg["parallelism"], "L_pq": cfg["L_pq"],
                     "recall_at_10": _r(cfg["recall_at_10"]), "target_recall": cfg["target_recall"],
                     "distinct_query_batches": cfg["distinct_query_batches"], "mean_evals_per_query": _r(cfg["mean_evals_per_query"], 6),
                     "mean_hops": _r(cfg["mean_hops"], 5), "setup_seconds": {k: _r(v, 3) for k, v in cfg["setup_seconds"].items()}}
    rf = line["roofline"]
    out["roofline"] = pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_hbm_only", "frac_cache_served", "kernel",
                                "kernel_ms_avg", "algorithmic_bytes_per_launch", "distinct_rows_frac", "frac_of_measured_stream_ceiling_6290"), 5)
    #  beside the headline's fraction -- which is CACHE-ASSISTED wherever few of a launch's row reads are first touches
    # (distinct_rows_frac) -- the two figures that are not: the fraction at L_pq 500 of the same index, and frac_hbm_only (random graph)
    p500 = line.get("L_pq_500")
    out["roofline"]["frac_at_L500"] = _r(p500["pct_of_8000"] / 100.0, 4) if p500 and p500.get("pct_of_8000") is not None else None
    out["roofline"]["frac_is"] = ("cache-assisted: %.1f %% of the launch's row reads are first touches of a row; where HBM alone must deliver: frac_hbm_only"
                                  % (100.0 * rf["distinct_rows_frac"])) if rf.get("distinct_rows_frac") is not None else None
    ts = rf.get("traffic_source")
    out["roofline"]["traffic_source"] = ts.split(" (")[0] if ts else None
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind", "host_cores", "L_pq", "value_without_prefetch", "gpu_over_cpu"), 5)
        smp = cb.get("sample") or ""
        out["cpu_baseline"]["sample"] = smp if len(smp) <= 200 else smp[:197] + "..."
    else:
        out["cpu_baseline"] = None
    c1 = line.get("cpu_baseline_1_thread")
    if c1:
        out["cpu_baseline_1_thread_qps"] = _r(c1.get("value"))
    c0 = line.get("cpu_baseline_config1")
    if c0:
        out["cpu_baseline_config1"] = pick(c0, ("value", "cores", "kind", "recall_at_10", "gpu_qps_same_inputs"))
    # sweep rows: [L_pq, QPS, recall@10, % of 8 TB/s]
    out["sweep_cols"] = ["L_pq", "qps", "recall_at_10", "pct_of_8000"]
    out["sweep"] = [[p["L_pq"], _r(p["qps"]), _r(p["recall_at_10"]), _r(p["pct_of_8000"], 3)] for p in line.get("L_pq_sweep") or []]
    w = line.get("roofline_worstcase")
    if w:
        out["worstcase"] = pick(w, ("qps", "frac", "kernel_ms_avg", "traffic"))
    g = line.get("gt_build")
    if g:
        out["gt_build"] = {"value": _r(g["value"]), "unit": "distances/s", "frac_of_mfma_peak": _r(g["roofline"]["frac"]),
                           "k2_resident_frac": _r((g.get("k2_device_resident") or {}).get("frac_of_mfma_peak")),
                           "cpu_value": _r((g.get("cpu_baseline") or {}).get("value"))}
        if g.get("k2_small_batch"):
            out["gt_build"]["k2_small_batch"] = g["k2_small_batch"]
        if g.get("k2_d512"):      # fractions of the fp32-MFMA peak at d = 512: {ip,l2}_{65536,10000}
            out["gt_build"]["k2_d512"] = g["k2_d512"]
    for name, key in (("two_streams_qps", "two_streams_pipelined"), ("host_form_qps", "host_form_pcie_inclusive")):
        if line.get(key):
            out[name] = _r(line[key].get("qps"))
    out["configs_summary"] = []
    for c in line.get("configs") or []:
        r_ = c.get("roofline") or {}
        out["configs_summary"].append({"name": c["name"], "nb": c.get("nb"), "dim": c.get("dim"), "L_pq": c["L_pq"], "qps": _r(c["value"]),
                                       "recall": _r(c["recall_at_k"]), "recall_k": c["recall_k"], "frac": _r(r_.get("frac")),
                                       "frac_hbm_only": _r(r_.get("frac_hbm_only")), "frac_cache_served": _r(r_.get("frac_cache_served")),
                                       "traffic": _r(r_.get("traffic")), "cpu_qps": _r((c.get("cpu_baseline") or {}).get("value")),
                                       "sweep": [[p["L_pq"], _r(p["pct_of_8000"], 3), _r(p["recall_at_k"], 3)] for p in c.get("L_pq_sweep") or []]})
    dm = line.get("device_memory")
    if dm:
        out["device_memory"] = dm
    out["full_record"] = full_paths
    return out
