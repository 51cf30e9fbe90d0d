#!/usr/bin/env python3
"""One fuzz case of tests/test_gpu_fuzz.py under several knob overrides, several times each: which form differs, and where."""
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as oracle  # noqa
from test_gpu_fuzz import make_case
from roargraph_amd import index as rg
seed = int(sys.argv[1])
base, q, off, nbrs, ep, metric, k, L, knobs = make_case(seed)
os.environ["RG_FORCE_CSR"] = str(knobs.pop("_csr"))
want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=2)
for over in ({}, {"visited_bytes": 0}, {"exact_filter": 0}, {"exact_filter": 0, "visited_bytes": 0}, {"lookahead": 2}, {"rows_per_pass": 16}, {"waves_per_cu": 0}):
    for rep in range(3):
        ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
        for name, v in {**knobs, **over}.items():
            ix.set(name, v)
        bad = []
        for call in range(3):
            got = ix.SearchRoarGraph(q, k, L)
            wrong = np.nonzero((got[0] != want[0]).any(1) | (got[3] != want[3]) | (got[2] != want[2]))[0]
            bad.append([int(x) for x in wrong[:6]])
            if len(wrong) and call == 0 and rep == 0:
                qi = int(wrong[0]); print("  q", qi, "got", got[0][qi], got[2][qi], got[3][qi], "want", want[0][qi], want[2][qi], want[3][qi])
        ix.close()
        print(json.dumps({"over": over, "rep": rep, "wrong_queries_per_call": bad}), flush=True)
