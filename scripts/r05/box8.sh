#!/bin/bash
# round 5, box 8: K2 selection with 512-entry candidate buffers (RG_GT_CAND=8) against 256
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box8
mkdir -p $OUT
cd $R
GT_FORMS="select256:;select512:RG_GT_CAND=8;sort512:RG_GT_CAND=8,RG_GT_DIAG=16;select256b:;select512b:RG_GT_CAND=8" timeout 900 python scripts/exp/gt_small_batch.py 200 10000000 8192,10000,16384,30000,65536,100000 > $OUT/gt_ab_cand.jsonl 2> $OUT/gt_ab.err
cat $OUT/gt_ab_cand.jsonl | cut -c1-200
tail -2 $OUT/gt_ab.err
