#!/bin/bash
# round 6, box 5: K2 at d = 512 with chunks of 128 floats (four barriers per tile) against 64; the reference's README workflow with the CLI twins
# on files at the t2i-10M shape; a plain lifecycle stress of 300 iterations at the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box5
mkdir -p $OUT
cd $R
export RG_FAULT_REPORT=$OUT/fault_report.txt
timeout 600 python -m pytest tests/test_gpu_groundtruth.py -m gpu -q > $OUT/pytest_gt.log 2>&1; echo "pytest gt rc=$?"; tail -2 $OUT/pytest_gt.log
for M in ip l2; do
  GT_FORMS="bk64:;bk128:RG_GT_BK=128;bk64_again:;bk128_again:RG_GT_BK=128" timeout 600 python scripts/exp/gt_small_batch.py 512 3000000 10000,30000,65536 $M > $OUT/gt_d512_bk_$M.jsonl 2> $OUT/gt_d512_bk_$M.err
  echo "gt d512 $M rc=$?"; cat $OUT/gt_d512_bk_$M.jsonl; tail -1 $OUT/gt_d512_bk_$M.err
done
timeout 900 python scripts/r06/fault_stress.py 300 1.0 0 $OUT/stress_300.json 2> $OUT/stress_300.err; echo "stress rc=$?"; tail -2 $OUT/stress_300.err; cat $OUT/stress_300.json
timeout 1800 bash scripts/e2e_cli_t2i10m.sh > $OUT/e2e_cli.log 2>&1; echo "e2e rc=$?"; tail -25 $OUT/e2e_cli.log; cp gpurun_out/e2e_cli/log.txt $OUT/e2e_cli_steps.txt 2>/dev/null; cp gpurun_out/e2e_cli/eval.csv $OUT/e2e_cli_eval.csv 2>/dev/null
