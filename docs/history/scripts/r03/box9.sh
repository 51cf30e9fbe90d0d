#!/bin/bash
# round 3, ninth box: exact visited words in uncached / fine-grained device memory (a test then costs a 32-byte sector instead of
# a 128-byte L2 line?)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box9
mkdir -p $OUT
cd $R
timeout 1500 python scripts/exp/k1_ab.py --L 300,500,700,1000,2000 --index-cache /tmp/ix.npz \
  --configs "words:visited=0,lookahead=0;words_uc:visited=0,lookahead=0,visited_uncached=1;look:visited=0,lookahead=1;look_uc:visited=0,lookahead=1,visited_uncached=1;look_fg:visited=0,lookahead=1,visited_uncached=2;look_uc_nofilter:visited=0,lookahead=1,visited_uncached=1,exact_filter=0;filter:visited=1;default:visited=2" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box9/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-18s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-18s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows), all(r["same_cmps"] in (None,True) for r in rows if r["config"]!="filter"))
PY
tail -3 $OUT/k1_ab.err
