#!/usr/bin/env python3
"""Calibration of FETCH_SIZE on the search kernel's own access pattern.  (GPU box, under rocprofv3)

K1b (rg_score_batch_dev: the gather + score routine of K1 without the traversal) scores N DISTINCT random rows of a
10M x 192 fp32 base -- rows of 768 bytes = six whole 128-byte lines at a 768-byte stride, the layout K1 reads with the
split rows -- each row exactly once, against a base far larger than the L2s and the Infinity Cache.  The bytes HBM must
deliver are therefore known: N x 768 (+ 4 N for the id list).  Run it under

    rocprofv3 --pmc FETCH_SIZE -d out -o s -- python scripts/exp/calib_fetch.py
    rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_HIT_sum -d out2 -o s -- python scripts/exp/calib_fetch.py

and compare rg_score_kernel's per-launch FETCH_SIZE (KiB) with `known_bytes_per_launch` printed here."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from roargraph_amd._lib import check, lib  # noqa: E402
from roargraph_amd.index import IndexBipartite  # noqa: E402

dev = torch.device("cuda", 0)
nb, d, n = 10_000_000, 192, 4_000_000
g = torch.Generator(device=dev); g.manual_seed(7)
base = torch.empty((nb, d), device=dev)
for s in range(0, nb, 1 << 20):
    base[s:s + (1 << 20)].normal_(generator=g)
off = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
nbrs = torch.zeros(1, dtype=torch.int32, device=dev)
ix = IndexBipartite.from_device(base, off, nbrs, 0, metric="ip")
q = torch.empty(d, device=dev).normal_(generator=g)
out = torch.zeros(n, device=dev)
st = torch.cuda.current_stream().cuda_stream
ms = []
for rep in range(6):
    ids = torch.randperm(nb, device=dev, generator=g)[:n].int().contiguous()      # distinct rows, another set every launch
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().rg_score_batch_dev(ix.handle, C.c_void_p(q.data_ptr()), C.c_void_p(ids.data_ptr()), C.c_uint32(n), C.c_void_p(out.data_ptr()), C.c_void_p(st)))
    e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
known = n * d * 4 + n * 4
print(json.dumps({"rows_per_launch": n, "row_bytes": d * 4, "known_bytes_per_launch": known, "known_KiB_per_launch": known / 1024.0,
                  "ms_per_launch": ms, "GBps": [known / (m / 1e3) / 1e9 for m in ms]}))
