#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point rg_search (queries up, results down, buffers allocated per call)
next to the device-resident rg_search_dev, bench workload."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from roargraph_amd.index import IndexBipartite
nb, dim, k, nq, deg, L = 10_000_000, 200, 10, 10000, 40, 500
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
base = torch.empty((nb, dim), device=dev)
for s in range(0, nb, 1 << 20):
    base[s:s + (1 << 20)].normal_(generator=g)
nbrs = torch.randint(0, nb, (nb * deg,), dtype=torch.int32, device=dev, generator=g)
off = torch.arange(0, nb + 1, dtype=torch.int64, device=dev) * deg
ix = IndexBipartite.from_device(base, off, nbrs, 0, metric="ip")
g.manual_seed(99)
q = torch.empty((nq, dim), device=dev).normal_(generator=g) * 0.5 + 0.3
hq = q.cpu().numpy()
ix.SearchRoarGraph(hq, k, L)
t0 = time.perf_counter()
for _ in range(5):
    ix.SearchRoarGraph(hq, k, L)
host_ms = (time.perf_counter() - t0) / 5 * 1e3
st = torch.cuda.current_stream().cuda_stream
ids = torch.zeros((nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((nq, k), device=dev)
cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
t0 = time.perf_counter()
for _ in range(5):
    ix.search_dev(q, k, L, ids, ds, cm, hp, stream=st)
torch.cuda.synchronize(); ix.search_wait(st)
dev_ms = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps({"rg_search_host_buffers_ms": round(host_ms, 2), "qps_pcie_inclusive": round(nq / host_ms * 1e3),
                  "rg_search_dev_ms": round(dev_ms, 2), "qps_device_resident": round(nq / dev_ms * 1e3)}))
