#!/usr/bin/env python3
"""Sweep launch knobs of the search kernel on the bench workload (same inputs as bench.py, built once)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from roargraph_amd.index import IndexBipartite

ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=200)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--deg", type=int, default=40)
ap.add_argument("--L", default="500")
ap.add_argument("--wpc", default="0")
ap.add_argument("--rpp", default="8")
ap.add_argument("--metric", default="ip")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--stride", type=int, default=0)
ap.add_argument("--diag", type=int, default=0)
ap.add_argument("--visited", default="0")
ap.add_argument("--filter", default="11")
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
stride = a.stride or a.dim
base = torch.zeros((a.nb, stride), device=dev)
for s in range(0, a.nb, 1 << 20):
    base[s:s + (1 << 20), :a.dim].normal_(generator=g)
nbrs = torch.randint(0, a.nb, (a.nb * a.deg,), dtype=torch.int32, device=dev, generator=g)
off = torch.arange(0, a.nb + 1, dtype=torch.int64, device=dev) * a.deg
g.manual_seed(99)
q = torch.empty((a.nq, a.dim), device=dev).normal_(generator=g) * 0.5 + 0.3
ix = IndexBipartite.from_device(base, off, nbrs, 0, metric=a.metric, dim=a.dim)
k = 10
ix.set('diag', a.diag)
ids = torch.zeros((a.nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((a.nq, k), device=dev)
cm = torch.zeros(a.nq, dtype=torch.int32, device=dev); hp = torch.zeros(a.nq, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for L in [int(x) for x in a.L.split(",")]:
    for wpc in [int(x) for x in a.wpc.split(",")]:
        for rpp in [int(x) for x in a.rpp.split(",")]:
          for vis in [int(x) for x in a.visited.split(",")]:
           for fl in ([int(x) for x in a.filter.split(",")] if vis else [0]):
            ix.set("waves_per_cu", wpc); ix.set("rows_per_pass", rpp); ix.set("visited", vis)
            if vis: ix.set("filter_log2", fl)
            ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st); torch.cuda.synchronize()
            best = 1e9
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            mc = cm.float().mean().item()
            print(json.dumps({"L": L, "wpc": wpc, "rpp": rpp, "vis": vis, "flt": fl, "ms": round(best, 3), "qps": round(a.nq / best * 1e3),
                              "evals": round(mc, 1), "GBps": round(a.nq * mc * 4 * a.dim / best / 1e6, 1)}), flush=True)
