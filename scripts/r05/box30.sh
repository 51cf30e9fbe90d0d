#!/bin/bash
# round 5, box 30: the new rule (32 rows in flight in the look-ahead form below L_pq 375) in the default mode; parity tests of the forms
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box30
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_baseline_shapes.py -x -q -k "not config3 and not config5_2p5m and not config4_10m" > $OUT/pytest.log 2>&1; grep -n "passed\|failed" $OUT/pytest.log | tail -2
timeout 2400 python scripts/exp/k1_ab.py --L 280,300,350,375,400 --nbatch 3 --reps 4 --index-cache /tmp/ix.npz \
  --configs "default:;forced16:rows_per_pass=16;default2:;look:visited=0,lookahead=1" > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python scripts/r05/ab_table.py $OUT/k1_ab.jsonl
