#!/usr/bin/env python3
"""A/B in one process: size of the LDS visited filter (rg_index_set "filter_log2"), bench workload, modes 2 and 1."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from roargraph_amd.index import IndexBipartite
nb, dim, k, nq, deg = 10_000_000, 200, 10, 10000, 40
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
base = torch.empty((nb, dim), device=dev)
for s in range(0, nb, 1 << 20):
    base[s:s + (1 << 20)].normal_(generator=g)
nbrs = torch.randint(0, nb, (nb * deg,), dtype=torch.int32, device=dev, generator=g)
off = torch.arange(0, nb + 1, dtype=torch.int64, device=dev) * deg
ix = IndexBipartite.from_device(base, off, nbrs, 0, metric="ip")
st = torch.cuda.current_stream().cuda_stream
g.manual_seed(99)
q = torch.empty((nq, dim), device=dev).normal_(generator=g) * 0.5 + 0.3
ids = torch.zeros((nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((nq, k), device=dev)
cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
for L in (500, 100, 1000):
    for vis in (2, 1):
        ix.set("visited", vis)
        row = {"L": L, "visited": vis}
        for rep in range(2):
            for f in (7, 8, 9, 10, 11):
                ix.set("filter_log2", f)
                ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(4):
                    ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st)
                b.record(); torch.cuda.synchronize(); ix.search_wait(st)
                row["f%d_%d" % (f, rep)] = round(nq / (a.elapsed_time(b) / 4) * 1e3)
                if vis == 1 and rep == 0: row["evals_f%d" % f] = round(float(cm.float().mean()))
        print(json.dumps(row), flush=True)
