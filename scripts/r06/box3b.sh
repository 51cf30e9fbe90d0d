#!/bin/bash
# round 6, box 3b: K2 after the cooperative re-score (L2) and the C = 0 first MFMA (IP): ground-truth tests, then the rates at d = 512 / 200
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box3b
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_groundtruth.py tests/test_gpu_baseline_shapes.py tests/test_prune_golden.py tests/test_gpu_cli.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest.log | tail -5
for M in ip l2; do
  GT_FORMS="default:;scalar_rescore:RG_GT_RESCORE_SCALAR=1" timeout 600 python scripts/exp/gt_small_batch.py 512 3000000 10000,30000,65536 $M > $OUT/gt_d512_$M.jsonl 2> $OUT/gt_d512_$M.err
  echo "gt d512 $M rc=$?"; cat $OUT/gt_d512_$M.jsonl; tail -1 $OUT/gt_d512_$M.err
done
GT_FORMS="default:" timeout 600 python scripts/exp/gt_small_batch.py 200 10000000 10000,30000,65536 ip > $OUT/gt_d200_ip.jsonl 2> $OUT/gt_d200_ip.err; cat $OUT/gt_d200_ip.jsonl
GT_FORMS="default:;scalar_rescore:RG_GT_RESCORE_SCALAR=1" timeout 600 python scripts/exp/gt_small_batch.py 200 10000000 10000,65536 l2 > $OUT/gt_d200_l2.jsonl 2> $OUT/gt_d200_l2.err; cat $OUT/gt_d200_l2.jsonl
