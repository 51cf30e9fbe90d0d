#!/usr/bin/env python3
"""One table (config x L_pq, % of 8 TB/s, and whether every output equalled the first configuration's) of a k1_ab.py record."""
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"config')]
Ls = sorted({r["L"] for r in rows}); cfgs = []
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-12s" % "config" + "  ".join("%6d" % L for L in Ls) + "  exact")
for c in cfgs:
    cell = lambda L: next(("%6.2f" % r["pct_of_8TBs"] for r in rows if r["config"] == c and r["L"] == L), "     -")
    print("%-12s" % c + "  ".join(cell(L) for L in Ls), " ", all(r["same_cmps"] in (None, True) and r["same_ids_hops"] in (None, True) for r in rows if r["config"] == c))
