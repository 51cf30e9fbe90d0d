#!/bin/bash
# round 3, sixth box: the second look-ahead form (exact next-pop prediction behind the scoring, rows up to 126 neighbours)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box6
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
tail -3 $OUT/tests_gpu.log
timeout 1500 python scripts/exp/k1_ab.py --L 300,500,700,1000,2000 --index-cache /tmp/ix.npz \
  --configs "words:visited=0,lookahead=0;look:visited=0,lookahead=1;look_gf1:visited=0,lookahead=1,gather_form=1;look_noguess_gf1:visited=0,lookahead=2,gather_form=1;look_gf1_nofilter:visited=0,lookahead=1,gather_form=1,exact_filter=0;look_gf1_r4:visited=0,lookahead=1,gather_form=1,rows_per_pass=16;filter_gf1:visited=1,gather_form=1;default_gf1:visited=2,gather_form=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box6/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-22s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-22s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows), all(r["same_cmps"] in (None,True) for r in rows if r["config"]!="filter_gf1"))
PY
tail -3 $OUT/k1_ab.err
