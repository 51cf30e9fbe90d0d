#!/usr/bin/env python3
"""End-to-end RoarGraph pipeline on synthetic cross-modal data (BASELINE config 5 shape: GPU ground truth -> CPU graph
build -> GPU search), everything through the C ABI:
   K2  train-query x base top-100          (compute_groundtruth step, README.md:62-75)
   rg_build_roargraph (CPU, T threads)     (test_build_roargraph step, README.md:79-97; M_sq=100 M_pjbp=35 L_pjpq=500)
   K2  test-query ground truth, K1 search sweep -> QPS @ recall@10
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from roargraph_amd import build, groundtruth, index, synth
from roargraph_amd.index import IndexBipartite

ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=1_000_000)
ap.add_argument("--ntrain", type=int, default=0, help="training queries (default = nb, as t2i: 10M base / 10M train)")
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--dim", type=int, default=200)
ap.add_argument("--metric", default="ip")
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--L", default="10,20,50,100,200,500,1000")
ap.add_argument("--out", default="")
ap.add_argument("--data", default="gaussian", help="gaussian | lowrank (roargraph_amd.synth.make_device_set)")
ap.add_argument("--rank", type=int, default=24)
ap.add_argument("--noise", type=float, default=0.05)
ap.add_argument("--build-device", type=int, default=-1, help=">= 0: phase 3 of the build on that GPU")
a = ap.parse_args()
ntrain = a.ntrain or a.nb
threads = a.threads or min(64, os.cpu_count() or 1)   # README.md:92-97 builds with T=64
dev = torch.device("cuda", 0)
base, train, q, desc = synth.make_device_set(dev, 1234, a.nb, ntrain, a.nq, a.dim, data=a.data, rank=a.rank, noise=a.noise)
st = torch.cuda.current_stream().cuda_stream
res = {"dataset": "%s, %s" % (desc, a.metric)}

t0 = time.perf_counter()
ti = torch.zeros((ntrain, 100), dtype=torch.int32, device=dev); tv = torch.zeros((ntrain, 100), device=dev)
groundtruth.gt_shard_dev(base, train, a.metric, 100, 0, ti, tv, stream=st); torch.cuda.synchronize()
res["gt_train_s"] = time.perf_counter() - t0
res["gt_train_distances_per_s"] = ntrain * a.nb / res["gt_train_s"]

hb = base.cpu().numpy()
t0 = time.perf_counter()
off, nbrs, ep = build.build_roargraph(hb, ti.cpu().numpy().view(np.uint32), a.metric, 100, 35, 500, num_threads=threads,
                                      device=a.build_device if a.build_device >= 0 else None)
res["build_s"] = time.perf_counter() - t0
res["build_threads"] = threads
res["build_phase3"] = "gpu" if a.build_device >= 0 else "cpu"
deg = np.diff(off.astype(np.int64))
res["degree_avg_min_max"] = [float(deg.mean()), int(deg.min()), int(deg.max())]

gi = torch.zeros((a.nq, 100), dtype=torch.int32, device=dev); gv = torch.zeros((a.nq, 100), device=dev)
groundtruth.gt_shard_dev(base, q, a.metric, 100, 0, gi, gv, stream=st); torch.cuda.synchronize()
gt = gi.cpu().numpy().view(np.uint32)
ix = IndexBipartite.from_device(base, torch.from_numpy(off.view(np.int64)).to(dev), torch.from_numpy(nbrs.view(np.int32)).to(dev),
                                ep, metric=a.metric)
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
    mc, mh = float(cm.float().mean()), float(hp.float().mean())
    rows.append({"L_pq": L, "qps": round(a.nq / best * 1e3), "recall_at_10": round(index.recall(ids.cpu().numpy().view(np.uint32), gt, k), 4),
                 "mean_evals": round(mc, 1), "mean_hops": round(mh, 1), "GBps": round(a.nq * mc * 4 * a.dim / best / 1e6, 1)})
    print(json.dumps(rows[-1]), flush=True)
res["curve"] = rows
print(json.dumps(res))
if a.out:
    open(a.out, "w").write(json.dumps(res, indent=1))
