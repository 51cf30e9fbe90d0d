#!/bin/bash
# round 3, twenty-eighth box: streamed gather (every register-set depth) against batch by batch, whole sweep
# parity, then A/B on the 10M index (the first configuration is exact: its cmps are the reference's)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box28
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
tail -4 $OUT/tests_gpu.log
timeout 1800 python scripts/exp/k1_ab.py --L 10,20,50,100,200,300,500,700,1000,1500,2000 --index-cache /tmp/ix.npz --reps 2 \
  --configs "words_batch:visited=0,lookahead=0,gather_roll=0;default_batch:visited=2,gather_roll=0;default:visited=2;filt_batch:visited=1,gather_roll=0;filt:visited=1;look_b_batch:visited=0,lookahead=1,gather_roll=0;look_b:visited=0,lookahead=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box28/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-14s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-14s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows), all(r["same_cmps"] in (None,True) for r in rows if not r["config"].startswith("filt")))
PY
tail -3 $OUT/k1_ab.err
