#!/usr/bin/env python3
"""One configuration, one allocation, many launches: does the byte-tag launch change its speed over TIME (clocks, throttling) or
only between allocations?  Prints the milliseconds of every launch and rocm-smi's clocks / power now and then."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from roargraph_amd import build, groundtruth, synth
from roargraph_amd.index import IndexBipartite
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nb, dim, nq, k = 10_000_000, 200, 10_000, 10
base, train, q0, _ = synth.make_device_set(dev, 1234, nb, nb // 5, nq, dim, data="lowrank", rank=32, q_seed=99)
cache = "/tmp/ix.npz"
if os.path.exists(cache):
    z = np.load(cache); h_off, h_nbrs, ep = z["off"], z["nbrs"], int(z["ep"])
else:
    ti, _ = groundtruth.groundtruth_distributed(base, 0, train, "ip", 100); torch.cuda.synchronize()
    h_off, h_nbrs, ep = build.build_roargraph(base.cpu().numpy(), ti.cpu().numpy().view(np.uint32), "ip", 100, 35, 500, num_threads=min(128, os.cpu_count() or 1), device=0)
    np.savez(cache, off=h_off, nbrs=h_nbrs, ep=ep)
del train
ix = IndexBipartite.from_device(base, torch.from_numpy(h_off.view(np.int64)).to(dev), torch.from_numpy(h_nbrs.view(np.int32)).to(dev), ep, metric="ip")
st = torch.cuda.current_stream().cuda_stream
qs = [q0, synth.make_device_set(dev, 1234, 1024, 0, nq, dim, data="lowrank", rank=32, q_seed=99 + 7919)[2]]
o = dict(ids=torch.zeros((nq, k), dtype=torch.int32, device=dev), ds=torch.zeros((nq, k), device=dev), cm=torch.zeros(nq, dtype=torch.int32, device=dev), hp=torch.zeros(nq, dtype=torch.int32, device=dev))
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        return [l.strip() for l in out.splitlines() if any(t in l for t in ("sclk", "mclk", "fclk", "Power", "junction", "memory", "Temperature"))][:12]
    except Exception as e:
        return [str(e)]
for name, knobs, L, n in (("look_b", {"visited": 0, "lookahead": 1}, 1000, 160), ("filt", {"visited": 1}, 1000, 60), ("look_b_again", {"visited": 0, "lookahead": 1}, 1000, 160),
                          ("look_b_2000", {"visited": 0, "lookahead": 1}, 2000, 80)):
    for kk, v in knobs.items():
        ix.set(kk, v)
    for b in range(2):
        ix.search_dev(qs[b], k, L, o["ids"], o["ds"], o["cm"], o["hp"], stream=st)
    ix.search_wait(st)
    print(json.dumps({"config": name, "L": L, "smi_before": smi()}), flush=True)
    ms = []
    t0 = time.time()
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ix.search_dev(qs[i & 1], k, L, o["ids"], o["ds"], o["cm"], o["hp"], stream=st); e1.record()
        ix.search_wait(st)
        ms.append(round(e0.elapsed_time(e1), 2))
    print(json.dumps({"config": name, "L": L, "seconds": round(time.time() - t0, 1), "ms": ms, "smi_after": smi()}), flush=True)
ix.close()
