#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE PMC passes of a bench workload into one search_traffic.json entry (what bench.py
reports as roofline.traffic).   usage: make_traffic_json.py <prefix> [<prefix> ...]  > search_traffic.json

For each prefix it reads <prefix>_fetch.pmc.json, <prefix>_write.pmc.json (scripts/rocprof_summary.py --json) and
<prefix>_trace.bench.json (the FULL bench record of the same command: bench.py --full-out), and emits {"workload": <the key bench.py
matches on: roofline.traffic_key>, ...}.  FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE is doubled on gfx950
(MI355X_MICROARCH.md, HBM section).  The figure is per launch of the timed search kernel -- the rg_search_kernel instantiation with
the most launches in the run; the profiled commands run one beam width only (no sweep) -- plus, in the filter + log form, the
rg_distinct_kernel launch that follows each of them."""
import json
import sys


def pick(d, needle, counter):
    best = None
    for k, v in d.items():
        if needle in k and counter in v and (best is None or v[counter]["launches"] > best[1]["launches"]):
            best = (k, v[counter])
    return best


out = []
for prefix in sys.argv[1:]:
    try:
        fetch = json.load(open(prefix + "_fetch.pmc.json"))
        write = json.load(open(prefix + "_write.pmc.json"))
        line = json.load(open(prefix + "_trace.bench.json"))
    except Exception as e:  # noqa: BLE001
        print("skipping %s: %r" % (prefix, e), file=sys.stderr)
        continue
    key = line["roofline"]["traffic_key"]
    fs, ws = pick(fetch, "rg_search_kernel", "FETCH_SIZE"), pick(write, "rg_search_kernel", "WRITE_SIZE")
    fb, wb, kern = 2.0 * 1024.0 * fs[1]["avg"], 1024.0 * ws[1]["avg"], [fs[0]]
    if "ELi1E" in fs[0].replace(" ", "") or ", 1, " in fs[0]:   # VIS = 1 instantiation: the id log is counted by K4 afterwards
        fd, wd = pick(fetch, "rg_distinct_kernel", "FETCH_SIZE"), pick(write, "rg_distinct_kernel", "WRITE_SIZE")
        if fd:
            fb += 2.0 * 1024.0 * fd[1]["avg"]
            kern.append(fd[0])
        if wd:
            wb += 1024.0 * wd[1]["avg"]
    alg = line["roofline"]["algorithmic_bytes_per_launch"]
    trace_ms = None      # average duration of that kernel in the --kernel-trace pass (the events of a bench run under rocprofv3 are inflated)
    try:
        short = kern[0].split("(")[0]
        for ln in open(prefix + "_trace.txt"):
            if ln.startswith(short[:60]) and "|" in ln:
                trace_ms = float(ln.split("|")[3])
                break
    except Exception:  # noqa: BLE001
        pass
    out.append({"workload": key, "kernels": kern, "fetch_bytes_corrected": fb, "write_bytes": wb,
                "fetch_launches": fs[1]["launches"], "write_launches": ws[1]["launches"],
                "algorithmic_bytes_per_launch": alg, "moved_over_algorithmic": (fb + wb) / alg if alg else None,
                "kernel_ms_avg_in_the_kernel_trace": trace_ms,
                "frac_of_8TBs_in_the_kernel_trace": alg / (trace_ms / 1e3) / 8e12 if trace_ms else None,
                "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the bench command "
                          "(scripts/profile_r05.sh); KiB units; FETCH_SIZE x2 (gfx950)"})
json.dump(out, sys.stdout, indent=1)
