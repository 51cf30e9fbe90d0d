#!/bin/bash
# round 6, box 2: (a) GPU suite at the new tree (prune goldens on the GPU kernel, K2 16-row form, bench split), default and RG_BALANCED_ALLOC=0;
# (b) K2 A/B at d = 512 (32-row against 16-row tiles, 256 against 384 keys; ip and l2) and d = 200 (16-row form with three workgroups per CU);
# (c) the driver's bench command; (d) rocprofv3 passes of the d = 200 random-graph launch in four row layouts / visited forms
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box2
mkdir -p $OUT
cd $R
export RG_FAULT_REPORT=$OUT/fault_report.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest default rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
RG_STRESS_ITERS=40 RG_BALANCED_ALLOC=0 timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_balanced_off.log 2>&1; echo "pytest RG_BALANCED_ALLOC=0 rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest_gpu_balanced_off.log | tail -8
for M in ip l2; do
  GT_FORMS="rs32_k256:RG_GT_RS16=0,RG_GT_CAND=4;rs32_k384:RG_GT_RS16=0,RG_GT_CAND=6;rs16_k256:RG_GT_CAND=4;rs16_k384:RG_GT_CAND=6;rs16_default:" \
    timeout 600 python scripts/exp/gt_small_batch.py 512 3000000 10000,30000,65536 $M > $OUT/gt_d512_$M.jsonl 2> $OUT/gt_d512_$M.err
  echo "gt d512 $M rc=$?"; cat $OUT/gt_d512_$M.jsonl; tail -2 $OUT/gt_d512_$M.err
done
GT_FORMS="default:;rs16:RG_GT_RS16=1" timeout 600 python scripts/exp/gt_small_batch.py 200 10000000 10000,30000,65536 ip > $OUT/gt_d200_ip.jsonl 2> $OUT/gt_d200_ip.err
echo "gt d200 rc=$?"; cat $OUT/gt_d200_ip.jsonl; tail -2 $OUT/gt_d200_ip.err
RG_BENCH_PROGRESS=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_default.json > $OUT/bench_default_stdout.txt 2> $OUT/bench_default_stderr.txt; echo "bench rc=$?"
tail -3 $OUT/bench_default_stderr.txt; cut -c1-1500 $OUT/bench_default_stdout.txt
OUT=$OUT/prof bash scripts/profile_r06.sh 2>&1 | tail -30
