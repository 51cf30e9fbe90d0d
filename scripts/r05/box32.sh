#!/bin/bash
# round 5, box 32: 16 against 32 rows in flight at the widest beams (the default takes 32 where LDS leaves eight residents or fewer)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box32
mkdir -p $OUT
cd $R
timeout 2400 python scripts/exp/k1_ab.py --L 1200,1500,1800,2000 --nbatch 3 --reps 3 --index-cache /tmp/ix.npz \
  --configs "default:visited=0,lookahead=1;r16:visited=0,lookahead=1,rows_per_pass=16;r32:visited=0,lookahead=1,rows_per_pass=32;default2:visited=0,lookahead=1;r16b:visited=0,lookahead=1,rows_per_pass=16" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python scripts/r05/ab_table.py $OUT/k1_ab.jsonl
