#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE PMC passes of a bench workload into one search_traffic.json entry (what bench.py
reports as roofline.traffic).   usage: make_traffic_json.py <prefix> [<prefix> ...]  > search_traffic.json

For each prefix it reads <prefix>_fetch.pmc.json, <prefix>_write.pmc.json (scripts/rocprof_summary.py --json) and
<prefix>_trace.bench.json (the bench line of the same command), and emits {"workload": <the key bench.py matches on>, ...}.
FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section).  The figure is
per launch of the timed search kernel -- the rg_search_kernel instantiation with the most launches in the run; the
profiled commands run one beam width only (no sweep) -- plus, in visited mode 2, the rg_distinct_kernel launch that
follows each of them."""
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
    fetch = json.load(open(prefix + "_fetch.pmc.json"))
    write = json.load(open(prefix + "_write.pmc.json"))
    line = json.loads(open(prefix + "_trace.bench.json").read().strip().splitlines()[-1])
    cfg = line["config"]
    wl = cfg["workload"]
    nb, dim = (int(x) for x in wl.split("base ")[1].split(" ")[0].split("x"))
    key = {"nb": nb, "dim": dim, "nq": int(wl.split(", ")[1].split(" queries")[0]), "k": int(wl.split("top-")[1].split(",")[0]),
           "metric": wl.split(" fp32 ")[1].split(",")[0], "data": "lowrank" if "low-rank" in wl else "gaussian",
           "graph": "roargraph" if "genuine RoarGraph" in wl else "random", "L": cfg["L_pq"], "visited": 2}
    fs, ws = pick(fetch, "rg_search_kernel", "FETCH_SIZE"), pick(write, "rg_search_kernel", "WRITE_SIZE")
    fb, wb, kern = 2.0 * 1024.0 * fs[1]["avg"], 1024.0 * ws[1]["avg"], [fs[0]]
    if "ELi1E" in fs[0].replace(" ", "") or ", 1, " in fs[0]:   # VIS = 1 instantiation: the id log is counted by K4 afterwards
        fd, wd = pick(fetch, "rg_distinct_kernel", "FETCH_SIZE"), pick(write, "rg_distinct_kernel", "WRITE_SIZE")
        if fd:
            fb += 2.0 * 1024.0 * fd[1]["avg"]
            kern.append(fd[0])
        if wd:
            wb += 1024.0 * wd[1]["avg"]
    out.append({"workload": key, "kernels": kern, "fetch_bytes_corrected": fb, "write_bytes": wb,
                "fetch_launches": fs[1]["launches"], "write_launches": ws[1]["launches"],
                "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
                "kernel_ms_avg_under_rocprof": line["roofline"]["kernel_ms_avg"],
                "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the bench command "
                          "(scripts/profile_r02.sh); KiB units; FETCH_SIZE x2 (gfx950)"})
json.dump(out, sys.stdout, indent=1)
