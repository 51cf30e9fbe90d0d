#!/bin/bash
# round 5, box 2: hub bitmap at wide beams -- fewer resident queries per CU (more LDS for the bitmap) and larger shares of the region
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box2
mkdir -p $OUT
cd $R
C="off8:visited=0,lookahead=1,hub_bits=0"
for w in 8 7 6 5 4; do for p in 60 90; do C="$C;w${w}p${p}:visited=0,lookahead=1,waves_per_cu=$w,hub_pct=$p,front_set=0"; done; done
timeout 2400 python scripts/exp/k1_ab.py --L 500,700,1000,1500,2000 --nbatch 3 --reps 2 --index-cache /tmp/ix.npz --configs "$C" > $OUT/k1_ab_hub_residents.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open('/root/repo/gpurun_out/r05_box2/k1_ab_hub_residents.jsonl') if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("config  "+"  ".join("%6d"%L for L in Ls))
for c in cfgs:
    print("%-7s "%c+"  ".join("%6.2f"%next(r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L) for L in Ls), all(r["same_cmps"] in (None,True) for r in rows if r["config"]==c))
PY
tail -3 $OUT/k1_ab.err
