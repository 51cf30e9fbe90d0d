#!/bin/bash
# LDS visited-filter size vs resident queries on the genuine 10M index (split rows): is a larger filter worth fewer waves?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/filt; mkdir -p $o
Ls=100,200,300,500,700,1000,2000
timeout 900 python scripts/exp/k1_phases.py --nb 10000000 --Ls $Ls --modes 1 --save /tmp/ix10 --out $o/f_auto.json > $o/f_auto.log 2>&1
for f in 12 13 14; do
  timeout 400 python scripts/exp/k1_phases.py --nb 10000000 --Ls $Ls --modes 1 --load /tmp/ix10 --set filter_log2=$f --out $o/f_$f.json > $o/f_$f.log 2>&1
done
python - <<'P'
import json
for f in ("auto","12","13","14"):
    try:
        d=json.load(open("gpurun_out/filt/f_%s.json"%f))
        print(f, " ".join("%d:%.1f%%(x%.2f)"%(r["L_pq"], r["alg_GBps"]/80, r["evals_performed"]/r["distinct_evals"]) for r in d["rows"]))
    except Exception as e: print(f,"failed",e)
P
