#!/bin/bash
# round 3, second box: bench.py's new flow (tests + the driver's command), and the per-phase cycle profile of the two
# exact-word forms of K1 on the genuine 10M index (instrumented build)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box2
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_cli.py -x -q -m gpu -k "bench" > $OUT/tests_bench.log 2>&1; echo "rc=$?" >> $OUT/tests_bench.log
tail -15 $OUT/tests_bench.log
RG_HIP_LIB=$R/roargraph_amd/librg_hip_prof.so timeout 900 python scripts/exp/k1_phases.py --nb 10000000 --Ls 500,1000,2000 --modes 0 --set lookahead=0 --save /tmp/ix --out $OUT/phases_atomics.json > $OUT/phases_atomics.log 2>&1
RG_HIP_LIB=$R/roargraph_amd/librg_hip_prof.so timeout 900 python scripts/exp/k1_phases.py --nb 10000000 --Ls 500,1000,2000 --modes 0 --set lookahead=1 --load /tmp/ix --out $OUT/phases_look.json > $OUT/phases_look.log 2>&1
python scripts/exp/show_phases.py $OUT/phases_atomics.json $OUT/phases_look.json
grep -h lookahead_hits $OUT/phases_look.json | head -3
(time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err) 2>&1 | tail -3
tail -3 $OUT/bench_default.err
python - <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r03_box2/bench_default.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("metric","value","ms_per_step","n_gpus")}); print(d["roofline"]); print(d["cpu_baseline"]); print(d["gt_build"])
    for p in d["L_pq_sweep"]: print(p["L_pq"], round(p["qps"]), round(p["recall_at_10"],4), round(p["pct_of_8000"],1), round(p["pct_of_6290"],1))
except Exception as e: print("no bench line:", e)
PY
