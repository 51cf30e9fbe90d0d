#!/bin/bash
# round 3, fifth box: more resident queries for the exact-words forms (no LDS filter, four register sets -> 128 VGPRs -> up to 16
# waves per CU where the beam leaves the LDS for it)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box5
mkdir -p $OUT
cd $R
timeout 1500 python scripts/exp/k1_ab.py --L 300,500,700,1000,2000 --index-cache /tmp/ix.npz \
  --configs "words:visited=0,lookahead=0;words_r4_nf:visited=0,lookahead=0,rows_per_pass=16,exact_filter=0;words_r4_nf_gf1:visited=0,lookahead=0,rows_per_pass=16,exact_filter=0,gather_form=1;nospec_r4_nf_gf1:visited=0,lookahead=2,rows_per_pass=16,exact_filter=0,gather_form=1;nospec_r4_nf_gf1_w12:visited=0,lookahead=2,rows_per_pass=16,exact_filter=0,gather_form=1,waves_per_cu=12;words_r4_f10_gf1:visited=0,lookahead=0,rows_per_pass=16,filter_log2=10,gather_form=1;filter_gf1:visited=1,gather_form=1;filter_gf1_r4_f11:visited=1,gather_form=1,rows_per_pass=16,filter_log2=11" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box5/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-22s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-22s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows))
PY
tail -3 $OUT/k1_ab.err
