#!/usr/bin/env python3
"""K1 row-stride experiment: the same 10M x 200 search with base rows at a padded stride (800 B rows straddle 128-B
lines; FETCH_SIZE is 13 % above the algorithmic bytes).  usage: stride_experiment.py [strides...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from roargraph_amd.index import IndexBipartite

nb, dim, nq, L, k, deg = 10_000_000, 200, 10_000, 500, 10, 40
strides = [int(a) for a in sys.argv[1:]] or [200, 208, 224, 256]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
dense = torch.empty((nb, dim), device=dev)
for s in range(0, nb, 1 << 20):
    dense[s:s + (1 << 20)].normal_(generator=g)
nbrs = torch.randint(0, nb, (nb * deg,), dtype=torch.int32, device=dev, generator=g)
off = torch.arange(0, nb + 1, dtype=torch.int64, device=dev) * deg
g.manual_seed(99)
q = torch.empty((nq, dim), device=dev).normal_(generator=g) * 0.5 + 0.3
ref = None
for st in strides:
    if st == dim:
        base = dense
    else:
        base = torch.zeros((nb, st), device=dev)
        base[:, :dim] = dense
    ix = IndexBipartite.from_device(base, off, nbrs, 0, metric="ip", dim=dim)
    ix.set("filter_log2", 9)
    ids = torch.zeros((nq, k), dtype=torch.int32, device=dev); dists = torch.zeros((nq, k), device=dev)
    cmps = torch.zeros(nq, dtype=torch.int32, device=dev); hops = torch.zeros(nq, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    out = {"stride_floats": st, "row_bytes": st * 4}
    for mode in (1, 2):
        ix.set("visited", mode)
        for _ in range(2):
            ix.search_dev(q, k, L, ids, dists, cmps, hops, stream=stream)
        ix.search_wait(stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ix.search_dev(q, k, L, ids, dists, cmps, hops, stream=stream)
        b.record(); torch.cuda.synchronize(); ix.search_wait(stream)
        ms = a.elapsed_time(b) / 5
        out["mode%d_ms" % mode] = ms
        out["mode%d_qps" % mode] = nq / ms * 1e3
    if ref is None:
        ref = (ids.clone(), dists.clone(), cmps.clone())
    else:
        assert torch.equal(ids, ref[0]) and torch.equal(dists.view(torch.int32), ref[1].view(torch.int32)) and torch.equal(cmps, ref[2])
    out["alg_GBps_mode2"] = float(cmps.sum().item()) * dim * 4 / (out["mode2_ms"] / 1e3) / 1e9
    print(json.dumps(out), flush=True)
    ix.close(); del ix
    if st != dim:
        del base
