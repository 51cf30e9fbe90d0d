#!/bin/bash
# round 4, box 4: is "conflict" an equivalence relation, and how many classes are there?  pairwise matrix over run representatives
# (2-GiB chunks in allocation order; box 3 saw A x14, B x4, A x12, B x24, A x16, B x58), then compositions at equal footprints;
# then the tail-count A/B with the knob reset fixed
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box4
mkdir -p $OUT
cd $R
timeout 900 scripts/exp/bin/alloc_map3 2 132 \
  P 1,6,12,15,17,20,28,32,40,50,56,68,72,85,100,115,126 \
  M 4,5,6,7,8,9/18,19,20,21,22,23 M 4,5,6,14,15,16/18,19,20,21,22,23 M 4,5,6,30,31,32/18,19,20,21,22,23 M 14,15,16,30,31,32/18,19,20,21,22,23 \
  M 4,5,14,15,30,31/18,19,20,21,22,23 M 4,5,14,15,70,71/18,19,20,21,22,23 M 4,14,30,54,70,100/18,19,20,21,22,23 \
  M 4,5,6,7,8,9/18,19,20,40,41,42 M 4,5,6,7,8,9/16,17,40,41,100,101 M 4,5,14,15,30,31/18,19,16,17,40,41 M 4,5,14,15,30,31/56,57,72,73,100,101 \
  > $OUT/alloc_map3.jsonl 2> $OUT/alloc_map3.err
cat $OUT/alloc_map3.jsonl | cut -c1-2500
timeout 1500 python scripts/exp/k1_ab.py --L 20,50,100,200,500,1000 --index-cache /tmp/ix.npz --pipelined \
  --configs "tail:visited=2;k4:visited=2,count_tail=0;filter:visited=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
tail -3 $OUT/k1_ab.err
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['config'], r['L'], r['pct_of_8TBs'], r['same_ids_hops'], r['same_cmps'])"
