#!/bin/bash
# round 4, box 27: K2 after the spill fix (candidate address and lane number made inside the rare paths: no scratch reload, no
# vmcnt(0) in front of a candidate store); wider candidate buffers (RG_GT_CAND=8); the balanced split in the one-round case
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box27
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_groundtruth.py -x -q -m gpu > $OUT/pytest_gt.log 2>&1; tail -3 $OUT/pytest_gt.log
GT_FORMS="default:;cand8:RG_GT_CAND=8;bal_one:RG_GT_BALANCE_ONE=1;bal_one_cand8:RG_GT_BALANCE_ONE=1,RG_GT_CAND=8;no_store:RG_GT_DIAG=8;no_epilogue:RG_GT_DIAG=2" \
  timeout 600 python scripts/exp/gt_small_batch.py 200 10000000 8192,10000,10880,32768,65536,100000 > $OUT/gt_fill_K100.jsonl 2> $OUT/gt_fill_K100.err
cat $OUT/gt_fill_K100.jsonl
GT_K=10 GT_FORMS="default:;bal_one:RG_GT_BALANCE_ONE=1" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 8192,10000,65536 > $OUT/gt_fill_K10.jsonl 2> $OUT/gt_fill_K10.err
cat $OUT/gt_fill_K10.jsonl
GT_FORMS="default:;bal_one:RG_GT_BALANCE_ONE=1" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 10000,65536 l2 > $OUT/gt_fill_l2.jsonl 2> $OUT/gt_fill_l2.err
cat $OUT/gt_fill_l2.jsonl
GT_FORMS="default:;bal_one:RG_GT_BALANCE_ONE=1" timeout 300 python scripts/exp/gt_small_batch.py 512 4000000 10000,65536 l2 > $OUT/gt_fill_d512_l2.jsonl 2> $OUT/gt_fill_d512_l2.err
cat $OUT/gt_fill_d512_l2.jsonl
tail -2 $OUT/*.err
