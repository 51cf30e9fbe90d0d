#!/usr/bin/env python3
"""K1b roofline microbench: random-row gather + score (no graph, no visited, no queue) over a large base."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from roargraph_amd.index import IndexBipartite
ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=200)
ap.add_argument("--n", type=int, default=50_000_000)
ap.add_argument("--metric", default="ip")
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
base = torch.empty((a.nb, a.dim), device=dev)
for s in range(0, a.nb, 1 << 20):
    base[s:s + (1 << 20)].normal_(generator=g)
off = torch.zeros(a.nb + 1, dtype=torch.int64, device=dev)
nbrs = torch.zeros(1, dtype=torch.int32, device=dev)
ix = IndexBipartite.from_device(base, off, nbrs, 0, metric=a.metric)
ids = torch.randint(0, a.nb, (a.n,), dtype=torch.int32, device=dev, generator=g)
q = torch.empty(a.dim, device=dev).normal_(generator=g)
out = torch.zeros(a.n, device=dev)
st = torch.cuda.current_stream().cuda_stream
ix.score_batch_dev(q, ids, out, stream=st); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ix.score_batch_dev(q, ids, out, stream=st); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
chk = (base[ids[:1000].long()] * q).sum(1)
err = (chk + out[:1000]).abs().max().item() if a.metric == "ip" else 0.0
print(json.dumps({"kernel": "rg_score_kernel", "nb": a.nb, "dim": a.dim, "n": a.n, "ms": round(best, 3),
                  "rows_per_s": round(a.n / best * 1e3), "GBps": round(a.n * 4 * a.dim / best / 1e6, 1), "check_err": err}))
