#!/bin/bash
# split-row layout: parity subset, then the bench with and without it on the same cached 10M index
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/split; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fast_mode.py -x -q -m gpu 2>&1 | tail -8 > $out/tests.txt
cat $out/tests.txt
common="--steps 20 --warmup 5 --index-cache /tmp/idx10m --gt-nq 0 --config1-nb 0 --cpu-seconds 0"
timeout 1200 python bench.py $common > $out/on.json 2> $out/on.err; echo "on rc=$?"
RG_SPLIT_ROWS=0 timeout 1200 python bench.py $common > $out/off.json 2> $out/off.err; echo "off rc=$?"
python - <<'P'
import json
for n in ("on","off"):
    try:
        b=json.loads(open(f"gpurun_out/split/{n}.json").read().strip().splitlines()[-1])
        print(n, round(b["value"]), b["config"]["L_pq"], round(b["roofline"]["frac"],4), [(p["L_pq"],round(p["pct_of_8000"],1)) for p in b["L_pq_sweep"]], round(b["roofline_worstcase"]["frac"],3))
    except Exception as e: print(n,"failed",e)
P
tail -3 $out/on.err $out/off.err
