#!/bin/bash
# round 5, box 12: K2 candidate buffers of 256 / 384 / 512 keys with the selection (no in-loop spill any more), new balance rule
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box12
mkdir -p $OUT
cd $R
GT_FORMS="c256:;c384:RG_GT_CAND=6;c512:RG_GT_CAND=8;c256b:;c384b:RG_GT_CAND=6;c512b:RG_GT_CAND=8" timeout 900 python scripts/exp/gt_small_batch.py 200 10000000 2048,8192,10000,16384,30000,50000,65536,100000 > $OUT/gt_ab_cand.jsonl 2> $OUT/gt.err
cut -c1-190 $OUT/gt_ab_cand.jsonl
tail -2 $OUT/gt.err
RG_GT_CAND=6 timeout 900 python -m pytest tests/test_gpu_groundtruth.py -x -q > $OUT/pytest_gt_c384.log 2>&1; tail -2 $OUT/pytest_gt_c384.log
