#!/bin/bash
# round 5, box 29: look-ahead tag form at L_pq 300 - 500 with 16 (default there) or 32 rows in flight per query, five repetitions
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box29
mkdir -p $OUT
cd $R
timeout 2400 python scripts/exp/k1_ab.py --L 300,350,400,450,500 --nbatch 3 --reps 5 --index-cache /tmp/ix.npz \
  --configs "r16:visited=0,lookahead=1;r32:visited=0,lookahead=1,rows_per_pass=32;r16b:visited=0,lookahead=1;r32b:visited=0,lookahead=1,rows_per_pass=32;r32_nofs:visited=0,lookahead=1,rows_per_pass=32,front_set=0" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python scripts/r05/ab_table.py $OUT/k1_ab.jsonl
