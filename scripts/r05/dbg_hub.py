import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import small_set
from oracle import pyoracle as po
from roargraph_amd import index as rg
base, q, off, nbrs, ep = small_set("ip", 4000, 200)
ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip")
ix.set("visited", 0); ix.set("lookahead", 1)
for L in (700, 2000):
    want = po.search(base, "ip", off, nbrs, ep, q, 10, L, nthreads=4)
    for fs in (0, -1):
        ix.set("front_set", fs)
        for rpp in (8, 16, 32):
            ix.set("rows_per_pass", rpp)
            for hb in (0, 8, 9, 10, 11, 12, 13, 14, 15, -1):
                ix.set("hub_bits", hb)
                got = ix.SearchRoarGraph(q, 10, L)
                d = got[2].astype(np.int64) - want[2].astype(np.int64)
                print("L", L, "fs", fs, "rpp", rpp, "hub_bits", hb, "m", ix.stat("hub_m_last"), "cmps diff: min", d.min(), "max", d.max(), "nonzero", int((d != 0).sum()),
                      "ids ok", bool((got[0] == want[0]).all()), "hops ok", bool((got[3] == want[3]).all()), flush=True)
