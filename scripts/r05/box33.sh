#!/bin/bash
# round 5, box 33: L_pq 200 - 290: the exact LDS set (with the tags behind it from 230) against the look-ahead tags with hubs and 32 rows in flight
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box33
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "exact_set or lset or default_mode" 2>&1 | tail -1
timeout 2400 python scripts/exp/k1_ab.py --L 240,260,270,280,290,300 --nbatch 3 --reps 4 --index-cache /tmp/ix.npz \
  --configs "default:;look:visited=0,lookahead=1;tags2:lset_tags=2;default2:" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python scripts/r05/ab_table.py $OUT/k1_ab.jsonl
