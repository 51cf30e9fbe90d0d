#!/usr/bin/env python3
"""LDS visited-filter size on a GENUINE RoarGraph index (structured 2M x 200 set, built here): QPS and evaluations
performed (re-scored nodes included) per filter size, visited modes 1 and 2.
What it showed: repeats are spread over ALL scored ids (two-entry recency sets or remembering only beam entrants change
nothing), so only capacity helps, and capacity costs resident queries: hence the automatic size in launch_k1."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from roargraph_amd import build, groundtruth, synth
from roargraph_amd.index import IndexBipartite
data = sys.argv[1] if len(sys.argv) > 1 else "lowrank"
nb, ntrain, nq, dim, k = 2_000_000, 400_000, 10000, 200, 10
dev = torch.device("cuda", 0)
base, train, q, desc = synth.make_device_set(dev, 1234, nb, ntrain, nq, dim, data=data, rank=32)
st = torch.cuda.current_stream().cuda_stream
ti = torch.zeros((ntrain, 100), dtype=torch.int32, device=dev); tv = torch.zeros((ntrain, 100), device=dev)
groundtruth.gt_shard_dev(base, train, "ip", 100, 0, ti, tv, stream=st); torch.cuda.synchronize()
off, nbrs, ep = build.build_roargraph(base.cpu().numpy(), ti.cpu().numpy().view(np.uint32), "ip", 100, 35, 500, num_threads=128, device=0)
ix = IndexBipartite.from_device(base, torch.from_numpy(off.view(np.int64)).to(dev), torch.from_numpy(nbrs.view(np.int32)).to(dev), ep, metric="ip")
print(json.dumps({"dataset": desc, "avg_degree": float(nbrs.size) / nb}), flush=True)
ids = torch.zeros((nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((nq, k), device=dev)
cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
for L in (100, 500):
    ix.set("visited", 0); ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
    distinct = float(cm.float().mean())
    for vis in (1, 2):
        ix.set("visited", vis)
        row = {"L": L, "visited": vis, "distinct_evals": round(distinct)}
        for f in (0, 9, 10, 11, 12):   # 0 = the library's automatic choice
            ix.set("filter_log2", f)
            f = "auto" if f == 0 else "%d" % f
            ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
            if vis == 1: row["performed_f%s" % f] = round(float(cm.float().mean()))
            best = 0
            for rep in range(2):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(4):
                    ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st)
                b.record(); torch.cuda.synchronize(); ix.search_wait(st)
                best = max(best, round(nq / (a.elapsed_time(b) / 4) * 1e3))
            row["qps_f%s" % f] = best
        print(json.dumps(row), flush=True)
