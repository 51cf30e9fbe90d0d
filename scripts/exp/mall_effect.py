#!/usr/bin/env python3
"""What the 256-MiB Infinity Cache is worth to a random row gather.  (GPU box)

K1b (rg_score_batch_dev: K1's gather + score without the traversal) reads 4,000,000 random 768-byte rows per launch from a
10M x 192 base; the rows are drawn (with repeats) from the first H rows only, H = 2^16 ... 10M, i.e. from a hot set of
H x 768 bytes.  No counter of this rocprofv3 separates Infinity-Cache hits from HBM reads (FETCH_SIZE counts both), so the
rate itself is the evidence: hot sets that fit the L2s (32 MiB) or the Infinity Cache (256 MiB) are served faster than
HBM can serve random rows, and a launch of the search whose reads go mostly to a quarter-gigabyte of rows
(bench.py: roofline.cache_served_frac_ceiling) sits between the two."""
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
rows = []
for H in (1 << 14, 1 << 16, 1 << 17, 1 << 18, 349525, 1 << 19, 1 << 20, 1 << 21, 1 << 22, nb):
    ms = []
    for rep in range(6):
        ids = torch.randint(0, H, (n,), device=dev, generator=g, dtype=torch.int64).int().contiguous()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().rg_score_batch_dev(ix.handle, C.c_void_p(q.data_ptr()), C.c_void_p(ids.data_ptr()), C.c_uint32(n), C.c_void_p(out.data_ptr()), C.c_void_p(st)))
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    m = sorted(ms)[len(ms) // 2]
    row = {"hot_rows": H, "hot_set_MiB": round(H * d * 4 / 2**20, 1), "ms_median": round(m, 4), "TBps_of_row_bytes": round(n * d * 4 / (m / 1e3) / 1e12, 2)}
    rows.append(row)
    print(json.dumps(row), flush=True)
