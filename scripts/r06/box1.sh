#!/bin/bash
# round 6, box 1: (a) GPU suite, default and with the class-balanced allocator OFF (kill switch);
# (b) lifecycle stress with the fault report on, light and heavy; (c) K2 at d = 512: where the time goes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box1
mkdir -p $OUT
cd $R
export RG_FAULT_REPORT=$OUT/fault_report.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest default rc=$?"; tail -3 $OUT/pytest_gpu.log
RG_STRESS_ITERS=30 RG_BALANCED_ALLOC=0 timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_balanced_off.log 2>&1; echo "pytest RG_BALANCED_ALLOC=0 rc=$?"; tail -3 $OUT/pytest_gpu_balanced_off.log
timeout 900 python scripts/r06/fault_stress.py 150 1.0 8 $OUT/stress_heavy.json 2> $OUT/stress_heavy.err; echo "stress heavy rc=$?"; tail -2 $OUT/stress_heavy.err
RG_MEM_CACHE_GIB=0 timeout 600 python scripts/r06/fault_stress.py 60 1.0 4 $OUT/stress_nocache.json 2> $OUT/stress_nocache.err; echo "stress no-cache rc=$?"; tail -2 $OUT/stress_nocache.err
ls -la $OUT/*fault_report* 2>/dev/null
# K2 d = 512: phase split by ablation (DIAG 2 = no filter / top-K, 1 = no base streaming), margin of the L2 re-score, 384-key buffers
for M in ip l2; do
  GT_FORMS="default:;nofilter:RG_GT_DIAG=2;nostream:RG_GT_DIAG=1;neither:RG_GT_DIAG=3;cand6:RG_GT_CAND=6;nomargin:RG_GT_NO_MARGIN=1" \
    timeout 600 python scripts/exp/gt_small_batch.py 512 3000000 10000,65536 $M > $OUT/gt_d512_$M.jsonl 2> $OUT/gt_d512_$M.err
  echo "gt d512 $M rc=$?"; cat $OUT/gt_d512_$M.jsonl
done
