#!/usr/bin/env python3
"""QPS @ recall@10 curve on a NAVIGABLE graph, everything built on the GPU with the product's own kernels.

No real index can be downloaded or (yet) built RoarGraph-style, so this check uses an exact k-NN graph:
  base  : mixture-of-Gaussians, nb x dim (default 1M x 200)
  graph : exact 32-NN lists from K2 (queries = base, K = 33, self dropped) + 4 random out-edges per node
  truth : exact top-100 of the test queries from K2
  search: K1 over an L_pq sweep; recall@10 with the reference's definition (rg_recall)
It shows that the search kernel delivers real recall on a navigable graph and what QPS that costs; the 10M-node
throughput number of bench.py uses a random graph (identical memory pattern, no meaningful recall).
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from roargraph_amd import groundtruth, index
from roargraph_amd.index import IndexBipartite

ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=200)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--metric", default="l2")
ap.add_argument("--clusters", type=int, default=2000)
ap.add_argument("--L", default="10,20,30,50,100,200,500")
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
centers = torch.empty((a.clusters, a.dim), device=dev).normal_(generator=g)
lab = torch.randint(0, a.clusters, (a.nb,), device=dev, generator=g)
base = centers[lab] + 0.6 * torch.empty((a.nb, a.dim), device=dev).normal_(generator=g)
qlab = torch.randint(0, a.clusters, (a.nq,), device=dev, generator=g)
q = centers[qlab] + 0.8 * torch.empty((a.nq, a.dim), device=dev).normal_(generator=g)
st = torch.cuda.current_stream().cuda_stream

t0 = time.time()
K = 33
kid = torch.zeros((a.nb, K), dtype=torch.int32, device=dev); kv = torch.zeros((a.nb, K), device=dev)
groundtruth.gt_shard_dev(base, base, a.metric, K, 0, kid, kv, stream=st); torch.cuda.synchronize()
t_knn = time.time() - t0
self_first = float((kid[:, 0].long() == torch.arange(a.nb, device=dev)).float().mean())
rnd = torch.randint(0, a.nb, (a.nb, 4), dtype=torch.int32, device=dev, generator=g)
nbrs = torch.cat([kid[:, 1:], rnd], dim=1).contiguous()
deg = nbrs.shape[1]
off = torch.arange(0, a.nb + 1, dtype=torch.int64, device=dev) * deg
ep = int(((base - base.mean(0)) ** 2).sum(1).argmin())
gt_i = torch.zeros((a.nq, 100), dtype=torch.int32, device=dev); gt_v = torch.zeros((a.nq, 100), device=dev)
groundtruth.gt_shard_dev(base, q, a.metric, 100, 0, gt_i, gt_v, stream=st); torch.cuda.synchronize()
gt = gt_i.cpu().numpy().view(np.uint32)

ix = IndexBipartite.from_device(base, off, nbrs.view(-1), ep, metric=a.metric)
k = 10
ids = torch.zeros((a.nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((a.nq, k), device=dev)
cm = torch.zeros(a.nq, dtype=torch.int32, device=dev); hp = torch.zeros(a.nq, dtype=torch.int32, device=dev)
rows = []
for L in [int(x) for x in a.L.split(",")]:
    ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st); e1.record(); ix.search_wait(st)
        best = min(best, e0.elapsed_time(e1))
    rec = index.recall(ids.cpu().numpy().view(np.uint32), gt, k)
    mc, mh = cm.float().mean().item(), hp.float().mean().item()
    rows.append({"L_pq": L, "qps": round(a.nq / best * 1e3), "recall_at_10": round(rec, 4), "mean_evals": round(mc, 1),
                 "mean_hops": round(mh, 1), "evals_per_hop": round(mc / mh, 1), "GBps": round(a.nq * mc * 4 * a.dim / best / 1e6, 1)})
    print(json.dumps(rows[-1]), flush=True)
res = {"dataset": "mixture of %d Gaussians, %d x %d, %s" % (a.clusters, a.nb, a.dim, a.metric), "graph": "exact 32-NN (K2) + 4 random, degree %d" % deg,
       "knn_build_s": round(t_knn, 2), "knn_distances_per_s": a.nb * a.nb / t_knn, "self_is_rank0": self_first, "curve": rows}
print(json.dumps(res))
if a.out:
    open(a.out, "w").write(json.dumps(res, indent=1))
