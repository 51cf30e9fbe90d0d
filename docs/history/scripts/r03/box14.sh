#!/bin/bash
# round 3, fourteenth box: where the id-log store of a hop is issued (behind the row loads vs after the scoring)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box14
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
tail -3 $OUT/tests_gpu.log
timeout 1500 python scripts/exp/k1_ab.py --L 20,50,100,200,300,500,700,1000 --index-cache /tmp/ix.npz --reps 4 \
  --configs "words:visited=0;late:visited=2,log_early=0;early:visited=2,log_early=1;filter_only:visited=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box14/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-14s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-14s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows), all(r["same_cmps"] in (None,True) for r in rows if r["config"]!="filter_only"))
PY
tail -3 $OUT/k1_ab.err
