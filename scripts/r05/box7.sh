#!/bin/bash
# round 5, box 7: K2 selection against the sort (one process per d), and the L_pq 300 - 400 forms of K1 in the default mode
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box7
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_groundtruth.py -x -q > $OUT/pytest_gt.log 2>&1; tail -3 $OUT/pytest_gt.log
GT_FORMS="select:;sort:RG_GT_DIAG=16;select2:;sort2:RG_GT_DIAG=16" timeout 900 python scripts/exp/gt_small_batch.py 200 10000000 8192,10000,30000,65536,100000 > $OUT/gt_ab_select.jsonl 2> $OUT/gt_ab.err
cat $OUT/gt_ab_select.jsonl
GT_FORMS="select:;sort:RG_GT_DIAG=16" timeout 900 python scripts/exp/gt_small_batch.py 512 3000000 10000,65536 l2 > $OUT/gt_ab_select_512.jsonl 2>> $OUT/gt_ab.err
cat $OUT/gt_ab_select_512.jsonl
timeout 1500 python scripts/exp/k1_ab.py --L 250,300,350,400,450 --nbatch 3 --reps 3 --index-cache /tmp/ix.npz \
  --configs "default:;nohub:hub_bits=0;look_p60:visited=0,lookahead=1,hub_pct=60;look_p90:visited=0,lookahead=1,hub_pct=90;look_p60_nofs:visited=0,lookahead=1,hub_pct=60,front_set=0;look_p90_nofs:visited=0,lookahead=1,hub_pct=90,front_set=0;look_nohub:visited=0,lookahead=1,hub_bits=0;lsettags:lset_tags=2" \
  > $OUT/k1_ab_300.jsonl 2> $OUT/k1_ab.err
python scripts/r05/ab_table.py $OUT/k1_ab_300.jsonl
tail -2 $OUT/k1_ab.err
