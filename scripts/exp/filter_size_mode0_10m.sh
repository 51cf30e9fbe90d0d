#!/bin/bash
# exact-words form (visited 0): size of the LDS filter that screens the atomics vs resident queries, wide beams, 10M index
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/filt0; mkdir -p $o
Ls=500,700,1000,1500,2000
timeout 900 python scripts/exp/k1_phases.py --nb 10000000 --Ls $Ls --modes 0 --save /tmp/ix10 --out $o/f_auto.json > $o/f_auto.log 2>&1
for f in 9 10 11 12; do
  timeout 400 python scripts/exp/k1_phases.py --nb 10000000 --Ls $Ls --modes 0 --load /tmp/ix10 --set filter_log2=$f --out $o/f_$f.json > $o/f_$f.log 2>&1
done
timeout 400 python scripts/exp/k1_phases.py --nb 10000000 --Ls $Ls --modes 0 --load /tmp/ix10 --set exact_filter=0 --out $o/f_none.json > $o/f_none.log 2>&1
python - <<'P'
import json
for f in ("auto","9","10","11","12","none"):
    try:
        d=json.load(open("gpurun_out/filt0/f_%s.json"%f))
        print(f, " ".join("%d:%.1f%%"%(r["L_pq"], r["alg_GBps"]/80) for r in d["rows"]))
    except Exception as e: print(f,"failed",e)
P
