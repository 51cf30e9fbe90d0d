#!/bin/bash
# round 3, nineteenth box: LDS visited filter of any size (filter_fill): parity, then A/B on the 10M index against the
# power-of-two sizes, with the resident count traded for filter entries (waves_per_cu 10 / 9 / 8)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box19
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
tail -4 $OUT/tests_gpu.log
timeout 1500 python scripts/exp/k1_ab.py --L 100,200,300,500,700,1000 --index-cache /tmp/ix.npz --reps 3 \
  --configs "f0:visited=1,filter_fill=0;f1:visited=1;f1w10:visited=1,waves_per_cu=10;f1w9:visited=1,waves_per_cu=9;f1w8:visited=1,waves_per_cu=8;f1w8r:visited=1,waves_per_cu=8,rows_per_pass=32;f1w7r:visited=1,waves_per_cu=7,rows_per_pass=32;d0:visited=2,filter_fill=0;d1:visited=2" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box19/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-14s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-14s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows), all(r["same_cmps"] in (None,True) for r in rows if not r["config"].startswith("f")))
PY
tail -3 $OUT/k1_ab.err
