#!/usr/bin/env python3
"""Small batches (online serving): latency of a batch of nq queries at L_pq = 100 / 500 with the automatic rows-in-flight
choice against forced 4 / 8 / 16 rows per pass.  Random out-degree-40 graph on the 10M x 200 base."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from roargraph_amd.index import IndexBipartite
nb, dim, k, deg = 10_000_000, 200, 10, 40
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
base = torch.empty((nb, dim), device=dev)
for s in range(0, nb, 1 << 20):
    base[s:s + (1 << 20)].normal_(generator=g)
nbrs = torch.randint(0, nb, (nb * deg,), dtype=torch.int32, device=dev, generator=g)
off = torch.arange(0, nb + 1, dtype=torch.int64, device=dev) * deg
ix = IndexBipartite.from_device(base, off, nbrs, 0, metric="ip")
st = torch.cuda.current_stream().cuda_stream
for nq in (1, 64, 512, 1536, 4096):
    g.manual_seed(99)
    q = torch.empty((nq, dim), device=dev).normal_(generator=g) * 0.5 + 0.3
    ids = torch.zeros((nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((nq, k), device=dev)
    cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
    for L in (100, 500):
        row = {"nq": nq, "L": L}
        for rpp in (0, 4, 8, 16):
            ix.set("rows_per_pass", rpp)
            ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st)
            b.record(); torch.cuda.synchronize(); ix.search_wait(st)
            row["ms_rpp%s" % ("auto" if rpp == 0 else rpp)] = round(a.elapsed_time(b) / 5, 3)
        print(json.dumps(row), flush=True)
