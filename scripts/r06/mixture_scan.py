#!/usr/bin/env python3
"""Which parameters of the 'mixture' synthetic family (roargraph_amd/synth.py) give an index like a real one (degree ~ 40, recall 0.9 at a
three-digit beam) with LOW reuse between queries?  1M x 200 IP, 200k training queries, M_sq=100 M_pjbp=35 L_pjpq=500: average degree,
recall@10 / evaluations / % of 8 TB/s per L_pq, share of a launch's row reads that are first touches."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from roargraph_amd import build, groundtruth, synth
from roargraph_amd.index import IndexBipartite, recall
dev = torch.device("cuda", 0)
nb, ntrain, nq, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 0, 10_000, 200
ntrain = nb // 5
st = torch.cuda.current_stream().cuda_stream
for spec in (sys.argv[2] if len(sys.argv) > 2 else "1000,0.35,0.1,0.45;1000,1.0,0.3,1.0;10000,0.5,0.2,0.6;100,0.7,0.3,0.8;1000,0.7,0.3,0.7").split(";"):
    os.environ["RG_MIXTURE"] = spec
    base, train, q, desc = synth.make_device_set(dev, 1234, nb, ntrain, nq, d, data="mixture", rank=128, q_seed=99)
    ti, _ = groundtruth.groundtruth_distributed(base, 0, train, "ip", 100)
    torch.cuda.synchronize()
    t0 = time.time()
    off, nbrs, ep = build.build_roargraph(synth.to_host(base), synth.to_host(ti).view(np.uint32), "ip", 100, 35, 500, num_threads=min(128, os.cpu_count() or 1), device=0)
    tb = time.time() - t0
    deg = np.diff(off.astype(np.int64))
    ix = IndexBipartite.from_device(base, torch.from_numpy(off.view(np.int64)).to(dev), torch.from_numpy(nbrs.view(np.int32)).to(dev), ep, metric="ip")
    gi = torch.zeros((nq, 100), dtype=torch.int32, device=dev); gv = torch.zeros((nq, 100), device=dev)
    groundtruth.gt_shard_dev(base, q, "ip", 100, 0, gi, gv, stream=st); torch.cuda.synchronize()
    gt = gi.cpu().numpy().view(np.uint32)
    rows = []
    for L in (50, 100, 200, 300, 500, 1000):
        ids = torch.zeros((nq, 10), dtype=torch.int32, device=dev); ds = torch.zeros((nq, 10), device=dev)
        cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
        for _ in range(2):
            ix.search_dev(q, 10, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ix.search_dev(q, 10, L, ids, ds, cm, hp, stream=st); ix.search_wait(st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        ev = float(cm.float().mean().item())
        rows.append({"L": L, "recall": round(recall(ids.cpu().numpy().view(np.uint32), gt, 10), 4), "evals": round(ev, 1), "hops": round(float(hp.float().mean().item()), 1),
                     "pct_of_8000": round(nq * ev * 800 / (ms / 1e3) / 8e12 * 100, 1)})
    print(json.dumps({"mixture": spec, "nb": nb, "avg_degree": round(float(deg.mean()), 2), "max_degree": int(deg.max()), "share_of_nodes_with_degree_le_2": round(float((deg <= 2).mean()), 3),
                      "build_s": round(tb, 1), "sweep": rows}), flush=True)
    ix.close()
    del base, train, q, ti, ix, gi, gv
    torch.cuda.empty_cache()
