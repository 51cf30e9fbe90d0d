#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE PMC passes of the default bench workload into the search_traffic.json that
bench.py reports as roofline.traffic.  usage: make_traffic_json.py <visited> <fetch.pmc.json> <write.pmc.json>

FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section).  The figure is
per launch of the timed search kernel (the instantiation with the most launches) plus, in visited mode 2, the
rg_distinct_kernel launch that follows each of them."""
import json
import sys


def pick(d, needle, counter):
    best = None
    for k, v in d.items():
        if needle in k and counter in v and (best is None or v[counter]["launches"] > best[1]["launches"]):
            best = (k, v[counter])
    return best


vis = int(sys.argv[1])
fetch = json.load(open(sys.argv[2]))
write = json.load(open(sys.argv[3]))
fs, ws = pick(fetch, "rg_search_kernel", "FETCH_SIZE"), pick(write, "rg_search_kernel", "WRITE_SIZE")
fb = 2.0 * 1024.0 * fs[1]["avg"]
wb = 1024.0 * ws[1]["avg"]
kern = [fs[0]]
if vis == 2:
    fd, wd = pick(fetch, "rg_distinct_kernel", "FETCH_SIZE"), pick(write, "rg_distinct_kernel", "WRITE_SIZE")
    if fd:
        fb += 2.0 * 1024.0 * fd[1]["avg"]
        kern.append(fd[0])
    if wd:
        wb += 1024.0 * wd[1]["avg"]
json.dump({"workload": {"nb": 10_000_000, "dim": 200, "nq": 10_000, "L": 500, "k": 10, "deg": 40, "metric": "ip",
                        "visited": vis, "real_index": False},
           "kernels": kern, "fetch_bytes_corrected": fb, "write_bytes": wb,
           "fetch_launches": fs[1]["launches"], "write_launches": ws[1]["launches"],
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                     "`bench.py --steps 5 --warmup 2 --cpu-seconds 0 --gt-nq 0 --recall-nb 0 --no-other-modes`; KiB units; FETCH_SIZE x2 (gfx950)"},
          sys.stdout, indent=1)
