#!/bin/bash
# round 3, twenty-sixth box: byte-tag look-ahead form, residents against screen size (explicit filter_log2 = no fill)
# parity, then A/B on the 10M index (the first configuration is exact: its cmps are the reference's)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box26
mkdir -p $OUT
cd $R
echo skip > $OUT/tests_gpu.log
tail -4 $OUT/tests_gpu.log
timeout 1800 python scripts/exp/k1_ab.py --L 300,500,700,1000,1500 --index-cache /tmp/ix.npz --reps 2 \
  --configs "look_b:visited=0,lookahead=1;look_b_f11:visited=0,lookahead=1,filter_log2=11;look_b_f10:visited=0,lookahead=1,filter_log2=10;look_b_f9:visited=0,lookahead=1,filter_log2=9;look_b_f9r32:visited=0,lookahead=1,filter_log2=9,rows_per_pass=32;filtlog:visited=2,lookahead=0" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box26/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-14s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-14s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows), all(r["same_cmps"] in (None,True) for r in rows if not r["config"].startswith("filt")))
PY
tail -3 $OUT/k1_ab.err
