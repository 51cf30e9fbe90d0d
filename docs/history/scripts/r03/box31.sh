#!/bin/bash
# round 3, thirty-first box: K4 with 16-byte log loads, 16 ids in flight per thread and the next round's loads ahead of the
# inserts -- parity, then the default mode (filter + log + K4) with the old and the new count in two processes on one box
# (batches enqueued back to back); the filter-only form beside each as the box's yardstick
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box31
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_baseline_shapes.py -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
tail -4 $OUT/tests_gpu.log
for V in oldk4 new; do
  LIB=$R/roargraph_amd/librg_hip.so; [ $V = oldk4 ] && LIB=$R/roargraph_amd/librg_hip_oldk4.so
  RG_HIP_LIB=$LIB timeout 1500 python scripts/exp/k1_ab.py --pipelined --nbatch 6 --L 20,50,100,300,500,700,1000 --index-cache /tmp/ix.npz --reps 3 \
    --configs "words:visited=0,lookahead=0;default:visited=2;filt:visited=1" > $OUT/k1_ab_$V.jsonl 2> $OUT/k1_ab_$V.err
done
python - <<'PY'
import json
for V in ("oldk4","new"):
    rows=[json.loads(l) for l in open("gpurun_out/r03_box31/k1_ab_%s.jsonl"%V) if l.startswith('{"config')]
    Ls=sorted({r["L"] for r in rows}); cfgs=[]
    for r in rows:
        if r["config"] not in cfgs: cfgs.append(r["config"])
    print(V); print("%-14s"%"config"+"".join("%9d"%L for L in Ls))
    for c in cfgs:
        print("%-14s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
    print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows), all(r["same_cmps"] in (None,True) for r in rows if not r["config"].startswith("filt")))
PY
tail -2 $OUT/k1_ab_new.err
