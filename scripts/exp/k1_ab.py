#!/usr/bin/env python3
"""A/B of K1 launch forms on the bench workload in ONE process on ONE box (box-to-box spread is +-3 %): the genuine
10M x 200 RoarGraph index of bench.py is built once (or loaded from --index-cache), then every configuration of --configs
is timed at every beam width of --L over --nbatch DISTINCT query batches (rotated, so no launch replays the previous one).

  --configs "name:knob=v,knob=v;name2:..."   knobs of rg_index_set
Output: one JSON line per (config, L): ms per batch, QPS, evaluations (the reference's cmps from the exact form), % of
8 TB/s, and whether ids/hops/cmps equal the first configuration's (they must: every form is exact)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from roargraph_amd import build, groundtruth, synth  # noqa: E402
from roargraph_amd.index import IndexBipartite  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=200)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--metric", default="ip")
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--rank", type=int, default=32)
ap.add_argument("--L", default="500,1000,2000")
ap.add_argument("--nbatch", type=int, default=3)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--index-cache", default="")
ap.add_argument("--pipelined", action="store_true")
ap.add_argument("--configs", default="atomics:visited=0,lookahead=0;look:visited=0,lookahead=1")
a = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ntrain = a.nb // 5
base, train, q0, desc = synth.make_device_set(dev, 1234, a.nb, ntrain, a.nq, a.dim, data="lowrank", rank=a.rank, q_seed=99)
t0 = time.time()
if a.index_cache and os.path.exists(a.index_cache):
    z = np.load(a.index_cache)
    h_off, h_nbrs, ep = z["off"], z["nbrs"], int(z["ep"])
else:
    ti, _ = groundtruth.groundtruth_distributed(base, 0, train, a.metric, 100)
    torch.cuda.synchronize()
    h_off, h_nbrs, ep = build.build_roargraph(base.cpu().numpy(), ti.cpu().numpy().view(np.uint32), a.metric, 100, 35, 500,
                                              num_threads=min(128, os.cpu_count() or 1), device=0)
    del ti
    if a.index_cache:
        np.savez(a.index_cache, off=h_off, nbrs=h_nbrs, ep=ep)
del train
print(json.dumps({"setup_s": round(time.time() - t0, 1), "edges": int(h_nbrs.size), "avg_deg": float(h_nbrs.size) / a.nb}), flush=True)
off = torch.from_numpy(h_off.view(np.int64)).to(dev)
nbrs = torch.from_numpy(h_nbrs.view(np.int32)).to(dev)
ix = IndexBipartite.from_device(base, off, nbrs, ep, metric=a.metric)
st = torch.cuda.current_stream().cuda_stream

# distinct query batches (the generator of bench.py: same mixing matrix, one seed per batch)
qs = [q0]
for b in range(1, a.nbatch):
    qs.append(synth.make_device_set(dev, 1234, 1024, 0, a.nq, a.dim, data="lowrank", rank=a.rank, q_seed=99 + 7919 * b)[2])
outs = [dict(ids=torch.zeros((a.nq, a.k), dtype=torch.int32, device=dev), ds=torch.zeros((a.nq, a.k), device=dev),
             cm=torch.zeros(a.nq, dtype=torch.int32, device=dev), hp=torch.zeros(a.nq, dtype=torch.int32, device=dev)) for _ in qs]


def run(b, L):
    o = outs[b]
    ix.search_dev(qs[b], a.k, L, o["ids"], o["ds"], o["cm"], o["hp"], stream=st)


configs = []
for c in a.configs.split(";"):
    name, _, kv = c.partition(":")
    configs.append((name, [(x.split("=")[0], int(x.split("=")[1])) for x in kv.split(",") if x]))
ALL_KNOBS = {"visited": 2, "lookahead": -1, "exact_filter": 1, "rows_per_pass": 0, "waves_per_cu": 0, "filter_log2": 0, "split_rows": 1,
             "gather_form": -1, "filter_min_indeg": 2, "visited_uncached": 0, "count_in_k1": -1, "log_early": 1, "shared_frontier": 0, "filter_fill": 1, "visited_bytes": -1, "gather_roll": 1, "lset": -1, "adaptive": 1, "lset_tags": 1, "front_set": -1, "hub_bits": -1, "hub_pct": -1}
ref = {}
for L in [int(x) for x in a.L.split(",")]:
    for name, kvs in configs:
        for kname, v in ALL_KNOBS.items():
            ix.set(kname, v)
        for kname, v in kvs:
            ix.set(kname, v)
        for b in range(len(qs)):          # settle (allocations, adaptive trials) on every batch once
            run(b, L)
        ix.search_wait(st)
        for b in range(len(qs)):
            run(b, L)
        ix.search_wait(st)
        ms = []
        if a.pipelined:     # the batches of a repetition enqueued back to back, one wait at the end (what bench.py times)
            for r in range(a.reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for b in range(len(qs)):
                    run(b, L)
                ix.search_wait(st)
                torch.cuda.synchronize()
                ms.append((time.perf_counter() - t0) * 1e3 / len(qs))
        else:
            for r in range(a.reps):
                for b in range(len(qs)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); run(b, L); e1.record()
                    ix.search_wait(st)
                    ms.append(e0.elapsed_time(e1))
        snap = [(o["ids"].clone(), o["hp"].clone(), o["cm"].clone()) for o in outs]
        same = None
        if L in ref:
            same = all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(snap, ref[L]))
            same_cmps = all(torch.equal(x[2], y[2]) for x, y in zip(snap, ref[L]))
        else:
            ref[L] = snap
            same_cmps = None
        evals = float(np.mean([r_[2].float().mean().item() for r_ in ref[L]]))
        m = float(np.mean(ms))
        print(json.dumps({"config": name, "L": L, "ms": round(m, 3), "ms_min": round(min(ms), 3), "qps": round(a.nq / m * 1e3),
                          "evals_ref": round(evals, 1), "pct_of_8TBs": round(a.nq * evals * 4 * a.dim / (m / 1e3) / 8e12 * 100, 2),
                          "same_ids_hops": same, "same_cmps": same_cmps}), flush=True)
ix.close()
