#!/bin/bash
# round 5, box 24: L_pq 300 - 450 of the look-ahead tag form: the exact set in front of the tags on / off, hub share 60 / 90 %, five repetitions
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box24
mkdir -p $OUT
cd $R
timeout 2400 python scripts/exp/k1_ab.py --L 300,350,400,450 --nbatch 3 --reps 5 --index-cache /tmp/ix.npz \
  --configs "fs_p60:visited=0,lookahead=1,hub_pct=60;nofs_p60:visited=0,lookahead=1,hub_pct=60,front_set=0;nofs_p90:visited=0,lookahead=1,hub_pct=90,front_set=0;fs_p90:visited=0,lookahead=1,hub_pct=90;fs_p60b:visited=0,lookahead=1,hub_pct=60;nofs_p60b:visited=0,lookahead=1,hub_pct=60,front_set=0;nofs_p90b:visited=0,lookahead=1,hub_pct=90,front_set=0;lsettags:lset_tags=2" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python scripts/r05/ab_table.py $OUT/k1_ab.jsonl
