#!/usr/bin/env python3
"""Helpers of scripts/first_8gpu.sh (never run on 8 GPUs: no round had such a node).
  make D nb nq dim     base.fbin / query.fbin of the bench's lowrank family under D
  check D n            n sampled rows of D/gt.bin against fp64 brute force on the GPU (ids equal outside fp64 tie bands of 1e-5)
  scale O              O/bench_n{1,2,4,8}.json -> O/SCALE.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main(argv):
    cmd = argv[0]
    if cmd == "make":
        import torch
        from roargraph_amd import io, synth
        D, nb, nq, dim = argv[1], int(argv[2]), int(argv[3]), int(argv[4])
        base, train, _, desc = synth.make_device_set(torch.device("cuda", 0), 1234, nb, nq, 16, dim, data="lowrank", rank=32)
        io.write_fbin(os.path.join(D, "base.fbin"), synth.to_host(base)); io.write_fbin(os.path.join(D, "query.fbin"), synth.to_host(train))
        print("written:", desc)
    elif cmd == "check":
        import torch
        from roargraph_amd import index as ix
        D, n = argv[1], int(argv[2])
        base, d = ix.fbin_load(os.path.join(D, "base.fbin")); q, _ = ix.fbin_load(os.path.join(D, "query.fbin"))
        ids, _ = ix.gt_load(os.path.join(D, "gt.bin"))
        rows = np.random.default_rng(0).choice(q.shape[0], n, replace=False)
        b = torch.from_numpy(base[:, :d]).cuda().double()
        s = torch.from_numpy(q[rows][:, :d]).cuda().double() @ b.T
        top = s.topk(ids.shape[1], dim=1)
        ref, val = top.indices.cpu().numpy(), top.values.cpu().numpy()
        same = ref == ids[rows]
        near = np.abs(np.diff(val, axis=1)) <= 1e-5 * (np.abs(val).max(axis=1, keepdims=True) + 1e-30)      # fp64 tie bands between neighbouring ranks
        tie = np.zeros_like(same)
        tie[:, 1:] |= near
        tie[:, :-1] |= near
        bad = int((~same & ~tie).sum())
        print(json.dumps({"rows_checked": n, "ids_equal_frac": float(same.mean()), "differences_outside_fp64_tie_bands": bad}))
        sys.exit(1 if bad else 0)
    elif cmd == "scale":
        O = argv[1]
        runs = []
        for n in (1, 2, 4, 8):
            p = os.path.join(O, "bench_n%d.json" % n)
            if os.path.exists(p):
                r = json.load(open(p))
                runs.append({"n_gpus": r["n_gpus"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                             "gt_build": (r.get("gt_build") or {}).get("value"), "roofline_frac": r["roofline"]["frac"]})
        json.dump({"what": "bench.py --gpus N, weak scaling (every rank its own 10,000-query batches; index replicated)", "runs": runs},
                  open(os.path.join(O, "SCALE.json"), "w"), indent=1)
        print(json.dumps(runs))
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main(sys.argv[1:])
