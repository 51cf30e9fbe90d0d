#!/bin/bash
# round 6, box 5: the tree with the arena -- GPU suite twice, the driver's bench command, smoke; K2 at d = 512 with chunks of 128 floats (four
# barriers per tile) against 64; 300 lifecycle iterations; the walk under address churn for 120 s; the reference's README workflow with the CLI twins
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box5
mkdir -p $OUT
cd $R
export RG_FAULT_REPORT=$OUT/fault_report.txt
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu_$i.log 2>&1; echo "pytest run $i rc=$?"; grep -E "passed|failed|^FAILED|Memory access" $OUT/pytest_gpu_$i.log | tail -4
done
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
RG_BENCH_PROGRESS=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_default.json > $OUT/bench_default_stdout.txt 2> $OUT/bench_default_stderr.txt; echo "bench rc=$?"; tail -2 $OUT/bench_default_stderr.txt; cut -c1-300 $OUT/bench_default_stdout.txt
for M in ip l2; do
  GT_FORMS="bk64:;bk128:RG_GT_BK=128;bk64_again:;bk128_again:RG_GT_BK=128" timeout 600 python scripts/exp/gt_small_batch.py 512 3000000 10000,30000,65536 $M > $OUT/gt_d512_bk_$M.jsonl 2> $OUT/gt_d512_bk_$M.err
  echo "gt d512 $M rc=$?"; cat $OUT/gt_d512_bk_$M.jsonl; tail -1 $OUT/gt_d512_bk_$M.err
done
timeout 900 python scripts/r06/fault_stress.py 300 1.0 0 $OUT/stress_300.json 2> $OUT/stress_300.err; echo "stress rc=$?"; tail -1 $OUT/stress_300.err; cat $OUT/stress_300.json
timeout 300 python scripts/r06/walk_stress.py 120 $OUT/walk_arena.json 2> $OUT/walk_arena.err; echo "walk (arena) rc=$?"; cat $OUT/walk_arena.json
timeout 1800 bash scripts/e2e_cli_t2i10m.sh > $OUT/e2e_cli.log 2>&1; echo "e2e rc=$?"; tail -22 $OUT/e2e_cli.log | cut -c1-220; cp gpurun_out/e2e_cli/log.txt $OUT/e2e_cli_steps.txt 2>/dev/null; cp gpurun_out/e2e_cli/eval.csv $OUT/e2e_cli_eval.csv 2>/dev/null
