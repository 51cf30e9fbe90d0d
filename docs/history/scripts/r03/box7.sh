#!/bin/bash
# round 3, seventh box: admission rule of the LDS visited filter by in-degree (threshold sweep), genuine 10M index
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box7
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
tail -3 $OUT/tests_gpu.log
C=""
for T in 0 2 3 4 6 8 12 16 24 32; do C="$C;f_T$T:visited=1,gather_form=1,filter_min_indeg=$T"; done
timeout 1500 python scripts/exp/k1_ab.py --L 100,300,500,700,1000,2000 --index-cache /tmp/ix.npz \
  --configs "words:visited=0,lookahead=0$C;w_T8:visited=0,lookahead=0,filter_min_indeg=8;look_T8:visited=0,lookahead=1,gather_form=1,filter_min_indeg=8" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box7/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-12s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-12s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows))
PY
tail -3 $OUT/k1_ab.err
