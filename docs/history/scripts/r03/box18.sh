#!/bin/bash
# round 3, eighteenth box (final HEAD record after the deterministic build): the round's defaults -- whole -m gpu suite, smoke, the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box18
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
tail -4 $OUT/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
(time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err) 2>&1 | tail -3
tail -3 $OUT/bench_default.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r03_box18/bench_default.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("metric","value","ms_per_step","n_gpus")}); r=d["roofline"]; print({k:r[k] for k in r if k not in ("reuse","replay_same_batch","traffic_source")})
    print(d["host_form_pcie_inclusive"]); print(d["two_streams_pipelined"]); print(d["roofline_worstcase"]["frac"]); print(d["cpu_baseline"]["value"], d["cpu_baseline"]["value_without_prefetch"])
    for p in d["L_pq_sweep"]: print(p["L_pq"], round(p["qps"]), round(p["recall_at_10"],4), round(p["pct_of_8000"],1))
except Exception as e: print("no bench line:", e)
PY
