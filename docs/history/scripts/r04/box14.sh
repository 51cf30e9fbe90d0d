#!/bin/bash
# round 4, box 14: why the exact LDS set does not pay at L_pq 60 with twelve and more residents (trace of the plan), rest of the suite
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box14
mkdir -p $OUT
cd $R
RG_TRACE_ADAPTIVE=1 timeout 900 python scripts/exp/k1_ab.py --L 50,60,80,100 --index-cache /tmp/ix.npz --pipelined --nbatch 4 \
  --configs "auto:visited=2;w12:visited=2,waves_per_cu=12;w11:visited=2,waves_per_cu=11" > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep "exact LDS set at" $OUT/k1_ab.err | sort | uniq -c | sort -rn | head -30
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['config'], r['L'], r['pct_of_8TBs'])"
( time timeout 3300 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_cli.py ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $OUT/pytest_cli.log 2>&1; tail -4 $OUT/pytest_cli.log
