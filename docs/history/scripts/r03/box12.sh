#!/bin/bash
# round 3, twelfth box: what a larger LDS visited filter is worth at EQUAL resident queries (the pay-off of moving the beam's ids
# out of the LDS, before building it)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box12
mkdir -p $OUT
cd $R
timeout 1500 python scripts/exp/k1_ab.py --L 500,1000,2000 --index-cache /tmp/ix.npz \
  --configs "words:visited=0;f12_w7:visited=1,filter_log2=12,waves_per_cu=7;f13_w7:visited=1,filter_log2=13,waves_per_cu=7;f12_w6:visited=1,filter_log2=12,waves_per_cu=6;f13_w6:visited=1,filter_log2=13,waves_per_cu=6;f14_w4:visited=1,filter_log2=14,waves_per_cu=4;f12_w4:visited=1,filter_log2=12,waves_per_cu=4;f9_w6:visited=1,filter_log2=9,waves_per_cu=6;f11_w6:visited=1,filter_log2=11,waves_per_cu=6;filter_auto:visited=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box12/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-14s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-14s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
PY
tail -3 $OUT/k1_ab.err
