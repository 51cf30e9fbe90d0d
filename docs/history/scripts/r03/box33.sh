#!/bin/bash
# round 3, thirty-third box: the byte-tag look-ahead form at d = 512 (webvid-2.5M shape, inner product): is it the better
# form of the exact set there too?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box33
mkdir -p $OUT
cd $R
timeout 1500 python scripts/exp/k1_ab.py --nb 2500000 --dim 512 --metric ip --L 50,100,200,500,1000,2000 --index-cache /tmp/ix512.npz --reps 2 \
  --configs "words:visited=0,lookahead=0;look_b:visited=0,lookahead=1;look_words:visited=0,lookahead=1,visited_bytes=0;filt:visited=1;default:visited=2;default_look:visited=2,lookahead=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box33/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-14s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-14s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows), all(r["same_cmps"] in (None,True) for r in rows if not r["config"].startswith("filt")))
PY
tail -3 $OUT/k1_ab.err
