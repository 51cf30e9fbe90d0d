#!/bin/bash
# round 4, box 7: the whole -m gpu suite, then the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box7
mkdir -p $OUT
cd $R
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
( time RG_TRACE_ALLOC=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r04_box7/bench.json") if x.startswith("{")]
if l:
    r=json.loads(l[-1])
    print("value", r["value"], "frac", r["roofline"]["frac"], "L", r["config"]["L_pq"], "recall", r["config"]["recall_at_10"], "total_s", r["config"]["setup_seconds"])
    print("forms", r["roofline"].get("kernel_forms_of_the_batches_so_far"), "mem", r.get("device_memory"))
    for p in r["L_pq_sweep"]: print(p["L_pq"], round(p["pct_of_8000"],1), round(p["recall_at_10"] or 0,4), round(p["mean_evals"]))
    print("worst", r["roofline_worstcase"]["frac"] if r["roofline_worstcase"] else None, "two", r["two_streams_pipelined"], "gt", r["gt_build"]["roofline"]["frac"] if r["gt_build"] else None)
    for c in r.get("configs") or []:
        print(c["name"], c["L_pq"], round(c["value"]), c["recall_at_k"], round(c["roofline"]["frac"],3), c["seconds"], (c["cpu_baseline"] or {}).get("value"), c["kernel_forms_of_the_batches"])
        print("   ", [(p["L_pq"], round(p["pct_of_8000"],1), round(p["recall_at_k"] or 0,3)) for p in c["L_pq_sweep"]])
PY
