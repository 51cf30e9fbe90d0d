#!/usr/bin/env python3
"""How well does in-degree predict which rows a launch reads?  (GPU box)  For the bench's 10M index: the share of a launch's row
reads (rg_search_reuse_stats) that go to the H nodes of highest IN-DEGREE, next to the share that goes to its H most READ rows.
If the two are close, a visited-word table indexed by in-degree rank would concentrate most tests in its first lines."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from roargraph_amd import build, groundtruth, synth
from roargraph_amd.index import IndexBipartite
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nb, d, nq = 10_000_000, 200, 10_000
base, train, q, _ = synth.make_device_set(dev, 1234, nb, nb // 5, nq, d, data="lowrank", rank=32, q_seed=99)
ti, _ = groundtruth.groundtruth_distributed(base, 0, train, "ip", 100); torch.cuda.synchronize()
h_off, h_nbrs, ep = build.build_roargraph(base.cpu().numpy(), ti.cpu().numpy().view(np.uint32), "ip", 100, 35, 500, num_threads=min(128, os.cpu_count() or 1), device=0)
del ti, train
off = torch.from_numpy(h_off.view(np.int64)).to(dev); nbrs = torch.from_numpy(h_nbrs.view(np.int32)).to(dev)
indeg = torch.bincount(nbrs.long(), minlength=nb)
by_indeg = torch.sort(indeg, descending=True, stable=True).indices
ix = IndexBipartite.from_device(base, off, nbrs, ep, metric="ip")
st = torch.cuda.current_stream().cuda_stream
ids = torch.zeros((nq, 10), dtype=torch.int32, device=dev); ds = torch.zeros((nq, 10), device=dev)
cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
ix.set("visited", 2); ix.set("count_in_k1", 0)
for L in (50, 500, 1000):
    ix.set("lookahead", 0)
    for _ in range(2):
        ix.search_dev(q, 10, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
    counts = torch.zeros(nb, dtype=torch.int32, device=dev)
    try:
        ev, dr = ix.reuse_stats(st, counts)
    except Exception as e:
        print(json.dumps({"L": L, "error": str(e)})); continue
    c = counts.double()
    by_reads = torch.cumsum(torch.sort(c, descending=True).values, 0) / ev
    by_in = torch.cumsum(c[by_indeg], 0) / ev
    print(json.dumps({"L": L, "reads": ev, "distinct": dr,
                      "share_to_top_H_by_reads": {str(h): round(float(by_reads[h - 1]), 4) for h in (16384, 131072, 349525, 1048576)},
                      "share_to_top_H_by_indegree": {str(h): round(float(by_in[h - 1]), 4) for h in (16384, 131072, 349525, 1048576, 2097152)}}), flush=True)
