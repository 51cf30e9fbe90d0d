#!/usr/bin/env python3
"""Visited modes 0 / 1 / 2 on a genuine RoarGraph index of a structured 2M x 200 set: where the lossy filter's repeats
cost more than the exact HBM words' atomics."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from roargraph_amd import build, groundtruth, synth
from roargraph_amd.index import IndexBipartite
nb, ntrain, nq, dim, k = 2_000_000, 400_000, 10000, 200, 10
dev = torch.device("cuda", 0)
base, train, q, desc = synth.make_device_set(dev, 1234, nb, ntrain, nq, dim, data="lowrank", rank=32)
st = torch.cuda.current_stream().cuda_stream
ti = torch.zeros((ntrain, 100), dtype=torch.int32, device=dev); tv = torch.zeros((ntrain, 100), device=dev)
groundtruth.gt_shard_dev(base, train, "ip", 100, 0, ti, tv, stream=st); torch.cuda.synchronize()
off, nbrs, ep = build.build_roargraph(base.cpu().numpy(), ti.cpu().numpy().view(np.uint32), "ip", 100, 35, 500, num_threads=128, device=0)
ix = IndexBipartite.from_device(base, torch.from_numpy(off.view(np.int64)).to(dev), torch.from_numpy(nbrs.view(np.int32)).to(dev), ep, metric="ip")
ids = torch.zeros((nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((nq, k), device=dev)
cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
for L in (100, 200, 500, 1000, 2000):
    row = {"L": L}
    for vis, name in ((0, "mode0_nofilter"), (0, "mode0"), (1, "mode1"), (2, "mode2")):
        ix.set("visited", vis); ix.set("exact_filter", 0 if name == "mode0_nofilter" else 1)
        ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
        if vis < 2: row["evals_%s" % name] = round(float(cm.float().mean()))
        best = 0
        for rep in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st)
            b.record(); torch.cuda.synchronize(); ix.search_wait(st)
            best = max(best, round(nq / (a.elapsed_time(b) / 3) * 1e3))
        row["qps_%s" % name] = best
    print(json.dumps(row), flush=True)
