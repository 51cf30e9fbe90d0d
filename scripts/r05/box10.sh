#!/bin/bash
# round 5, box 10: instrumented K2 (RG_GT_PROF=1) at 10,000 and 30,000 queries with and without the quota thresholds
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box10
mkdir -p $OUT
cd $R
GT_FORMS="quota:RG_GT_PROF=1;own:RG_GT_PROF=1,RG_GT_NOSHARE=1;nofilter:RG_GT_DIAG=2" timeout 900 python scripts/exp/gt_small_batch.py 200 10000000 10000,30000,65536 > $OUT/gt_prof.jsonl 2> $OUT/gt_prof.err
cut -c1-190 $OUT/gt_prof.jsonl
grep "rg_gt prof" $OUT/gt_prof.err
