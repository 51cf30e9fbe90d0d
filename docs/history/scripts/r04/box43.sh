#!/bin/bash
# round 4, box 43: K2 with the candidate counts in a register and compaction by the owning wave (RG_GT_RC=1, d = 200): parity suite, A/B, shares
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box43
mkdir -p $OUT
cd $R
RG_GT_RC=1 timeout 600 python -m pytest tests/test_gpu_groundtruth.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for rep in 1; do
  GT_FORMS="default:;rc:RG_GT_RC=1" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 8192,10000,30000,65536 2> $OUT/err$rep.txt | tee -a $OUT/gt_ab.jsonl
done
GT_FORMS="rc_prof:RG_GT_RC=1,RG_GT_PROF=1" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 8192,65536 2> $OUT/prof.txt > /dev/null; grep "rg_gt prof" $OUT/prof.txt | sort -u
