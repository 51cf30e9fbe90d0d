#!/bin/bash
# round 3, eleventh box: the distinct count inside K1 (parity, then A/B against K4 at narrow beams), then the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box11
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_concurrency.py tests/test_gpu_baseline_shapes.py -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
tail -4 $OUT/tests_gpu.log
timeout 1500 python scripts/exp/k1_ab.py --L 20,50,80,100,150,200 --index-cache /tmp/ix.npz --reps 5 \
  --configs "words:visited=0;k4:visited=2,count_in_k1=0;in_k1:visited=2;in_k1_300:visited=2,count_in_k1=300;filter_only:visited=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box11/k1_ab.jsonl") if l.startswith('{"config')]
Ls=sorted({r["L"] for r in rows}); cfgs=[]
for r in rows:
    if r["config"] not in cfgs: cfgs.append(r["config"])
print("%-14s"%"config"+"".join("%9d"%L for L in Ls))
for c in cfgs:
    print("%-14s"%c+"".join("%9.1f"%next((r["pct_of_8TBs"] for r in rows if r["config"]==c and r["L"]==L),0) for L in Ls))
print("all exact:", all(r["same_ids_hops"] in (None,True) for r in rows), all(r["same_cmps"] in (None,True) for r in rows if r["config"]!="filter_only"))
PY
tail -3 $OUT/k1_ab.err
(time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err) 2>&1 | tail -3
tail -3 $OUT/bench_default.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r03_box11/bench_default.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("metric","value","ms_per_step","n_gpus")}); r=d["roofline"]; print({k:r[k] for k in r if k not in ("replay_same_batch","traffic_source")})
    for p in d["L_pq_sweep"]: print(p["L_pq"], round(p["qps"]), round(p["recall_at_10"],4), round(p["pct_of_8000"],1))
except Exception as e: print("no bench line:", e)
PY
