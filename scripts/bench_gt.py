#!/usr/bin/env python3
"""Ground-truth (K2) throughput: distances/s and fp32-MFMA TFLOP/s of one GPU's shard pass (BASELINE metric #2)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from roargraph_amd import groundtruth
ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=10_000_000)
ap.add_argument("--nq", type=int, default=65_536)
ap.add_argument("--dim", type=int, default=200)
ap.add_argument("--K", type=int, default=100)
ap.add_argument("--metric", default="ip")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--no-warmup", action="store_true", help="single launch (for rocprofv3 kernel-trace averages)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
base = torch.empty((a.nb, a.dim), device=dev)
for s in range(0, a.nb, 1 << 20):
    base[s:s + (1 << 20)].normal_(generator=g)
q = torch.empty((a.nq, a.dim), device=dev).normal_(generator=g) * 0.5 + 0.3
ids = torch.zeros((a.nq, a.K), dtype=torch.int32, device=dev)
vals = torch.zeros((a.nq, a.K), device=dev)
st = torch.cuda.current_stream().cuda_stream
if not a.no_warmup:
    groundtruth.gt_shard_dev(base[: 1 << 16], q[:1024], a.metric, a.K, 0, ids[:1024], vals[:1024], stream=st); torch.cuda.synchronize()
best = 1e18
for _ in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); groundtruth.gt_shard_dev(base, q, a.metric, a.K, 0, ids, vals, stream=st); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
# spot check 4 queries against torch fp64
chk = (q[:4].double() @ base.double().T) if a.metric == "ip" else -torch.cdist(q[:4].double(), base.double()) ** 2
top = chk.topk(a.K, dim=1)
same = float((top.indices.int() == ids[:4]).float().mean())
dps = a.nq * a.nb / (best / 1e3)
print(json.dumps({"kernel": "rg_gt_kernel", "nb": a.nb, "nq": a.nq, "dim": a.dim, "K": a.K, "metric": a.metric, "ms": round(best, 2),
                  "distances_per_s": dps, "TFLOPs": 2 * a.dim * dps / 1e12, "frac_of_157.3": 2 * a.dim * dps / 157.3e12,
                  "top_ids_match_fp64": same}))
