#!/usr/bin/env python3
"""SURVEY 8 f-2, measured: does a hub-first physical layout help the search?  (GPU box)

The genuine 10M x 200 index of bench.py is built once (or loaded from --index-cache).  Layout B renumbers the nodes by
in-degree, most linked first -- base rows AND adjacency rows of the hubs then sit together at the start of both arrays
(same pages, same cache sets) -- which is what a hub-first layout inside the library would do with ids remapped on the way
in and out.  Both layouts are searched with the same distinct query batches at the same beam widths, in one process.
The graphs are isomorphic, so recall and the evaluation counts must agree up to (distance, id) tie order; the report is
QPS / % of 8 TB/s per layout.  With --pmc-tag the script only runs layout A or B (for a rocprofv3 counter pass per layout:
TCC_HIT_sum / TCC_MISS_sum, FETCH_SIZE)."""
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
from roargraph_amd import index as ixmod  # noqa: E402
from roargraph_amd.index import IndexBipartite  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=200)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--rank", type=int, default=32)
ap.add_argument("--L", default="50,500")
ap.add_argument("--nbatch", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--index-cache", default="")
ap.add_argument("--only", default="", help="A or B: run one layout only (counter passes)")
ap.add_argument("--knobs", default="", help="knob=value,... set on both indexes (e.g. visited=0,lookahead=1: the byte-tag exact set)")
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
    ti, _ = groundtruth.groundtruth_distributed(base, 0, train, "ip", 100)
    torch.cuda.synchronize()
    h_off, h_nbrs, ep = build.build_roargraph(base.cpu().numpy(), ti.cpu().numpy().view(np.uint32), "ip", 100, 35, 500,
                                              num_threads=min(128, os.cpu_count() or 1), device=0)
    del ti
    if a.index_cache:
        np.savez(a.index_cache, off=h_off, nbrs=h_nbrs, ep=ep)
del train
off = torch.from_numpy(h_off.view(np.int64)).to(dev)
nbrs = torch.from_numpy(h_nbrs.view(np.int32)).to(dev).long()
nd = a.nb
# ---- layout B: new id = rank by in-degree (stable: ties keep the old order)
indeg = torch.bincount(nbrs, minlength=nd)
order = torch.sort(indeg, descending=True, stable=True).indices           # order[new] = old
rank = torch.empty_like(order); rank[order] = torch.arange(nd, device=dev)
deg = off[1:] - off[:-1]
deg_b = deg[order]
off_b = torch.zeros(nd + 1, dtype=torch.int64, device=dev); off_b[1:] = torch.cumsum(deg_b, 0)
src = torch.repeat_interleave(torch.arange(nd, device=dev), deg)           # old source of every edge
pos = off_b[rank[src]] + (torch.arange(nbrs.numel(), device=dev) - off[src])
nbrs_b = torch.empty_like(nbrs); nbrs_b[pos] = rank[nbrs]
base_b = base[order].contiguous()
ep_b = int(rank[ep].item())
top = indeg[order[: nd // 100]].sum().item() / max(int(nbrs.numel()), 1)
print(json.dumps({"setup_s": round(time.time() - t0, 1), "avg_deg": nbrs.numel() / nd, "max_indeg": int(indeg.max().item()),
                  "share_of_edges_into_top_1pct": round(top, 4)}), flush=True)
del src, pos
st = torch.cuda.current_stream().cuda_stream
qs = [q0] + [synth.make_device_set(dev, 1234, 1024, 0, a.nq, a.dim, data="lowrank", rank=a.rank, q_seed=99 + 7919 * b)[2] for b in range(1, a.nbatch)]
gts = []
gi = torch.zeros((a.nq, 100), dtype=torch.int32, device=dev); gv = torch.zeros((a.nq, 100), device=dev)
for qb in qs:
    groundtruth.gt_shard_dev(base, qb, "ip", 100, 0, gi, gv, stream=st); torch.cuda.synchronize()
    gts.append(gi.cpu().numpy().view(np.uint32).copy())
k = 10
layouts = [("A_as_built", base, off, nbrs.int(), ep, None), ("B_hub_first", base_b, off_b, nbrs_b.int(), ep_b, order)]
if a.only:
    layouts = [l for l in layouts if l[0].startswith(a.only)]
res = {}
for name, b_, o_, n_, e_, back in layouts:
    ix = IndexBipartite.from_device(b_, o_, n_, e_, metric="ip")
    for kv in [x for x in a.knobs.split(",") if x]:
        ix.set(kv.split("=")[0], int(kv.split("=")[1]))
    outs = [dict(ids=torch.zeros((a.nq, k), dtype=torch.int32, device=dev), ds=torch.zeros((a.nq, k), device=dev),
                 cm=torch.zeros(a.nq, dtype=torch.int32, device=dev), hp=torch.zeros(a.nq, dtype=torch.int32, device=dev)) for _ in qs]
    for L in [int(x) for x in a.L.split(",")]:
        def run(b):
            o = outs[b]
            ix.search_dev(qs[b], k, L, o["ids"], o["ds"], o["cm"], o["hp"], stream=st)
        for _ in range(2):
            for b in range(len(qs)):
                run(b)
            ix.search_wait(st)
        ms = []
        for r in range(a.reps):
            for b in range(len(qs)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(b); e1.record(); ix.search_wait(st)
                ms.append(e0.elapsed_time(e1))
        evals = float(np.mean([o["cm"].float().mean().item() for o in outs]))
        rec = []
        for b, o in enumerate(outs):
            ids = o["ids"].long()
            if back is not None:
                ids = back[ids]
            rec.append(ixmod.recall(ids.cpu().numpy().astype(np.uint32), gts[b], 10))
        m = float(np.mean(ms))
        row = {"layout": name, "L": L, "ms": round(m, 3), "qps": round(a.nq / m * 1e3), "mean_evals": round(evals, 1),
               "recall_at_10": round(float(np.mean(rec)), 4), "pct_of_8TBs": round(a.nq * evals * 4 * a.dim / (m / 1e3) / 8e12 * 100, 2)}
        res[(name, L)] = row
        print(json.dumps(row), flush=True)
    ix.close()
