#!/usr/bin/env python3
"""K2 at query counts that do not fill the chip with whole query blocks: the balanced work split (round 4) against equal items
handed out by a counter (RG_GT_NOBALANCE=1).  One launch per (form, nq) over a 10M x d base, % of the 157.3 TFLOP/s fp32-MFMA peak."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from roargraph_amd import groundtruth
dev = torch.device("cuda", 0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
nqs = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "10000,16384,30000,65536,100000").split(",")]
metric = sys.argv[4] if len(sys.argv) > 4 else "ip"
g = torch.Generator(device=dev); g.manual_seed(1)
base = torch.empty((nb, d), device=dev)
for s in range(0, nb, 1 << 20):
    base[s:s + (1 << 20)].normal_(generator=g)
K = int(os.environ.get("GT_K", "100"))
for nq in nqs:
    q = torch.empty((nq, d), device=dev).normal_(generator=g) * 0.5 + 0.3
    ids = torch.zeros((nq, K), dtype=torch.int32, device=dev); vals = torch.zeros((nq, K), device=dev)
    ref = None
    forms = (("default", {}), ("equal_items", {"RG_GT_NOBALANCE": "1"}))
    if os.environ.get("GT_FORMS"):       # "default" / "equal_items", or "name:ENV=V,ENV=V;name2:..." (experiment switches of rg_gt.hip)
        spec = os.environ["GT_FORMS"]
        if ":" in spec:
            forms = tuple((f.partition(":")[0], dict(kv.split("=") for kv in f.partition(":")[2].split(",") if kv)) for f in spec.split(";"))
        else:
            forms = tuple(f for f in forms if f[0] in spec.split(","))
    for form, env in forms:
        for k_ in ("RG_GT_NOSHARE", "RG_GT_NOBALANCE", "RG_GT_CAND", "RG_GT_BALANCE_ONE", "RG_GT_DIAG", "RG_GT_PROF", "RG_GT_NO_MARGIN", "RG_GT_RESCORE_SCALAR"):
            os.environ.pop(k_, None)
        os.environ.update(env)
        groundtruth.gt_shard_dev(base, q, metric, K, 0, ids, vals); torch.cuda.synchronize()
        t0 = time.perf_counter()
        groundtruth.gt_shard_dev(base, q, metric, K, 0, ids, vals); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = None if ref is None else bool(torch.equal(ref, ids))
        ref = ids.clone() if ref is None else ref
        print(json.dumps({"d": d, "K": K, "nq": nq, "form": form, "seconds": round(dt, 4), "TFLOPs": round(2.0 * d * nq * nb / dt / 1e12, 2),
                          "frac_of_157.3": round(2.0 * d * nq * nb / dt / 1e12 / 157.3, 4), "ids_equal_first_form": same}), flush=True)
