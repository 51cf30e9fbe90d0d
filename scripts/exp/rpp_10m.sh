#!/bin/bash
# rows in flight per query (4 vs 8 register sets) at mid / wide beams, both visited forms, 10M index
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/rpp; mkdir -p $o
Ls=200,300,500,700,1000,1500,2000
timeout 900 python scripts/exp/k1_phases.py --nb 10000000 --Ls $Ls --modes 1,0 --save /tmp/ix10 --out $o/auto.json > $o/auto.log 2>&1
for r in 16 32; do
  timeout 500 python scripts/exp/k1_phases.py --nb 10000000 --Ls $Ls --modes 1,0 --load /tmp/ix10 --set rows_per_pass=$r --out $o/r_$r.json > $o/r_$r.log 2>&1
done
python - <<'P'
import json
for f in ("auto","r_16","r_32"):
    try:
        d=json.load(open("gpurun_out/rpp/%s.json"%f))
        for m in (1,0):
            print(f, "mode", m, " ".join("%d:%.1f%%"%(r["L_pq"], r["alg_GBps"]/80) for r in d["rows"] if r["visited"]==m))
    except Exception as e: print(f,"failed",e)
P
