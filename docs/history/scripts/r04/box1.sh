#!/bin/bash
# round 4, box 1: (a) does the allocation METHOD of the 19 GiB of visited tags pin the fast placement mode?
# (b) look-ahead byte-tag form: eight residents with 32 rows in flight + bit screen vs 12-16 residents with 16 rows and little/no screen
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box1
mkdir -p $OUT
cd $R
timeout 600 scripts/exp/bin/alloc_place 5 0 19 1 > $OUT/alloc_place_fresh.jsonl 2> $OUT/alloc_place_fresh.err
timeout 600 scripts/exp/bin/alloc_place 5 1 19 0 > $OUT/alloc_place_churn.jsonl 2> $OUT/alloc_place_churn.err
timeout 1500 python scripts/exp/k1_ab.py --L 500,700,1000 --index-cache /tmp/ix.npz \
  --configs "look8:visited=0;look_r4_w12:visited=0,rows_per_pass=16,waves_per_cu=12;look_r4_w14:visited=0,rows_per_pass=16,waves_per_cu=14;look_r4_w16_noscreen:visited=0,rows_per_pass=16,waves_per_cu=16,filter_log2=4;look_r8_noscreen:visited=0,filter_log2=4;filter:visited=1;default:visited=2" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
tail -3 $OUT/k1_ab.err
cat $OUT/alloc_place_fresh.jsonl $OUT/alloc_place_churn.jsonl | cut -c1-200
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['config'], r['L'], r['pct_of_8TBs'], r['same_ids_hops'], r['same_cmps'])"
