#!/bin/bash
# round 3, fourth box: is a wide-beam launch bound by the memory system?  (resident queries 8 / 6 / 4; plain-load tests without
# speculation; compute-layout gather) -- one process, genuine 10M index
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box4
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "exact_words or compute_layout" > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
tail -3 $OUT/tests_gpu.log
timeout 1500 python scripts/exp/k1_ab.py --L 500,1000,2000 --index-cache /tmp/ix.npz \
  --configs "words:visited=0,lookahead=0;words_w6:visited=0,lookahead=0,waves_per_cu=6;words_w4:visited=0,lookahead=0,waves_per_cu=4;nospec:visited=0,lookahead=2;nospec_gf1:visited=0,lookahead=2,gather_form=1;nospec_gf1_nofilter:visited=0,lookahead=2,gather_form=1,exact_filter=0;filter:visited=1;filter_gf1:visited=1,gather_form=1;filter_gf1_w6:visited=1,gather_form=1,waves_per_cu=6;filter_gf1_w12:visited=1,gather_form=1,waves_per_cu=12,rows_per_pass=16" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box4/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-22s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-22s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows))
PY
tail -3 $OUT/k1_ab.err
