#!/usr/bin/env python3
"""Where does a hop of K1 spend its time on a GENUINE RoarGraph index?  (GPU box; instrumented build `make prof`)

  RG_HIP_LIB=roargraph_amd/librg_hip_prof.so python scripts/exp/k1_phases.py --nb 2000000 --save /tmp/ix
  python scripts/exp/k1_phases.py --load /tmp/ix            # same index, product build: undisturbed QPS

Builds a structured (low-rank) set + its RoarGraph index with the product pipeline (K2 truth, GPU-assisted build), then
runs the search at several beam widths and visited modes.  With the instrumented library it prints, per (L_pq, mode),
the share of wave cycles per phase (pop / adjacency wait / visited filter / gather+score / merge / other), fresh
neighbours per hop, and how often a speculative expansion of the next-to-pop node would have been consumed."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=2_000_000)
    ap.add_argument("--dim", type=int, default=200)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--metric", default="ip")
    ap.add_argument("--data", default="lowrank")
    ap.add_argument("--rank", type=int, default=32)
    ap.add_argument("--Ls", default="100,500,1000,2000")
    ap.add_argument("--modes", default="1,0,2")
    ap.add_argument("--save", default="")
    ap.add_argument("--load", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--set", default="", help="comma list of knob=value applied to the index")
    args = ap.parse_args()
    import torch
    from roargraph_amd import build, groundtruth, synth, index as ixmod
    from roargraph_amd._lib import lib, LIB_PATH
    from roargraph_amd.index import IndexBipartite
    dev = torch.device("cuda", 0)
    prof = hasattr(lib(), "rg_prof_buffer")
    ntrain = args.nb // 5
    base, train, q, desc = synth.make_device_set(dev, 1234, args.nb, ntrain, args.nq, args.dim, data=args.data, rank=args.rank, q_seed=99)
    st = torch.cuda.current_stream().cuda_stream
    if args.load:
        z = np.load(args.load + ".npz")
        off, nbrs, ep = z["off"], z["nbrs"], int(z["ep"])
    else:
        ti = torch.zeros((ntrain, 100), dtype=torch.int32, device=dev); tv = torch.zeros((ntrain, 100), device=dev)
        t0 = time.time()
        groundtruth.gt_shard_dev(base, train, args.metric, 100, 0, ti, tv, stream=st); torch.cuda.synchronize()
        t1 = time.time()
        off, nbrs, ep = build.build_roargraph(base.cpu().numpy(), ti.cpu().numpy().view(np.uint32), args.metric, 100, 35, 500,
                                              num_threads=min(128, os.cpu_count() or 1), device=0)
        print("gt %.1fs build %.1fs avg deg %.1f" % (t1 - t0, time.time() - t1, nbrs.size / args.nb), flush=True)
        del ti, tv
        if args.save:
            np.savez(args.save + ".npz", off=off, nbrs=nbrs, ep=ep)
    del train
    gi = torch.zeros((args.nq, 100), dtype=torch.int32, device=dev); gv = torch.zeros((args.nq, 100), device=dev)
    groundtruth.gt_shard_dev(base, q, args.metric, 100, 0, gi, gv, stream=st); torch.cuda.synchronize()
    gt = gi.cpu().numpy().view(np.uint32)
    ix = IndexBipartite.from_device(base, torch.from_numpy(off.view(np.int64)).to(dev), torch.from_numpy(nbrs.view(np.int32)).to(dev),
                                    ep, metric=args.metric)
    for kv in [x for x in args.set.split(",") if x]:
        k_, v_ = kv.split("="); ix.set(k_, int(v_))
    ids = torch.zeros((args.nq, args.k), dtype=torch.int32, device=dev); ds = torch.zeros((args.nq, args.k), device=dev)
    cm = torch.zeros(args.nq, dtype=torch.int32, device=dev); hp = torch.zeros(args.nq, dtype=torch.int32, device=dev)
    pbuf = torch.zeros((args.nq, 24), dtype=torch.int64, device=dev)
    if prof:
        lib().rg_prof_buffer(C.c_void_p(pbuf.data_ptr()))
    rows = []
    names = ["pop", "adj_wait", "filter", "score_issue", "merge", "other", "gather_wait"]
    for L in [int(x) for x in args.Ls.split(",")]:
        ix.set("visited", 0); ix.search_dev(q, args.k, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
        distinct = float(cm.float().mean())
        for mode in [int(x) for x in args.modes.split(",")]:
            ix.set("visited", mode)
            for _ in range(2):
                ix.search_dev(q, args.k, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ix.search_dev(q, args.k, L, ids, ds, cm, hp, stream=st); b.record(); ix.search_wait(st)
            ms = a.elapsed_time(b)
            row = {"L_pq": L, "visited": mode, "qps": args.nq / (ms / 1e3), "ms": ms,
                   "recall_at_10": ixmod.recall(ids.cpu().numpy().view(np.uint32), gt, 10),
                   "distinct_evals": distinct, "evals_performed": float(cm.float().mean()), "hops": float(hp.float().mean()),
                   "alg_GBps": args.nq * distinct * 4 * args.dim / (ms / 1e3) / 1e9}
            if prof:
                p = pbuf.cpu().numpy().astype(np.float64)
                tot = p[:, :8].sum()
                row["phase_share"] = {names[i]: round(float(p[:, i].sum() / tot), 4) for i in range(7)}
                row["cycles_per_hop"] = float(tot / max(hp.float().sum().item(), 1))
                c = p[:, 8:16].sum(0)
                m = pbuf.cpu().numpy()[:, 16:20]
                row["merge_split"] = {"search_dedup": round(float(m[:, 0].sum() / tot), 4), "rank_cursor": round(float(m[:, 1].sum() / tot), 4), "shift": round(float(m[:, 2].sum() / tot), 4)}
                nm = max(float((m[:, 3] >> 32).sum()), 1.0)
                row["valid_cands_per_hop"] = float((m[:, 3] >> 32).sum() / max(hp.float().sum().item(), 1))
                row["chunks_moved_per_hop"] = float((m[:, 3] & 0xffffffff).sum() / max(hp.float().sum().item(), 1))
                row["chunks_per_hop"] = float(c[0] / max(hp.float().sum().item(), 1))
                row["fresh_per_hop"] = float(c[1] / max(hp.float().sum().item(), 1))
                row["spec_hit_rate"] = 0.0
                row["spec_tries_per_hop"] = float(c[2] / max(hp.float().sum().item(), 1))
                row["deg_per_hop"] = float(c[5] / max(hp.float().sum().item(), 1))
                row["lookahead_hits_per_hop"] = float(c[6] / max(hp.float().sum().item(), 1))   # VIS = 2: the early-fetched node was the one popped
            rows.append(row)
            print(json.dumps(row), flush=True)
    out = {"lib": os.path.basename(LIB_PATH), "dataset": desc, "nb": args.nb, "avg_degree": nbrs.size / args.nb, "rows": rows}
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
