#!/bin/bash
# round 5, box 4: the look-ahead form without the wait for the marks' acknowledgement (lookahead=3), A/B on the 10M bench index
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box4
mkdir -p $OUT
cd $R
timeout 2400 python scripts/exp/k1_ab.py --L 300,500,1000,2000 --nbatch 3 --reps 2 --index-cache /tmp/ix.npz --configs "look1:visited=0,lookahead=1;look3:visited=0,lookahead=3;look3w:visited=0,lookahead=3,visited_bytes=0;look1w:visited=0,lookahead=1,visited_bytes=0" > $OUT/k1_ab_noack.jsonl 2> $OUT/k1_ab.err
python scripts/r05/ab_table.py $OUT/k1_ab_noack.jsonl
tail -3 $OUT/k1_ab.err
