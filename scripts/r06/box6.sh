#!/bin/bash
# round 6, box 6: the final tree once more (GPU suite with -x as the driver runs it, smoke, the driver's bench command), K2's phase split at d = 200
# (RG_GT_PROF), then a larger sample of the walk under address churn: one long arena run, three runs with the address policy of rounds 4 - 5
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box6
mkdir -p $OUT
cd $R
RG_FAULT_REPORT=$OUT/fault_report.txt timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_default.json > $OUT/bench_default_stdout.txt 2> $OUT/bench_default_stderr.txt; echo "bench rc=$?"; tail -2 $OUT/bench_default_stderr.txt; cut -c1-260 $OUT/bench_default_stdout.txt
for NQ in 65536 10000; do
  GT_FORMS="prof:RG_GT_PROF=1" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 $NQ ip > $OUT/gt_prof_$NQ.jsonl 2> $OUT/gt_prof_$NQ.err; grep "rg_gt prof" $OUT/gt_prof_$NQ.err | tail -1 | cut -c1-400
done
timeout 700 python scripts/r06/walk_stress.py 540 $OUT/walk_arena_long.json 2> $OUT/walk_arena_long.err; echo "arena long rc=$?"; cat $OUT/walk_arena_long.json
for i in 3 4 5; do
  RG_MEM_VA=leak timeout 420 python scripts/r06/walk_stress.py 240 $OUT/walk_leak_$i.json 2> $OUT/walk_leak_$i.err; echo "leak run $i rc=$?"; grep -a "Memory access fault\|\[walk\]" $OUT/walk_leak_$i.err | tail -2 | cut -c1-200; cat $OUT/walk_leak_$i.json 2>/dev/null
done
timeout 300 python scripts/r06/walk_stress.py 60 $OUT/walk_arena_after.json 2> $OUT/walk_arena_after.err; echo "arena after rc=$?"; cat $OUT/walk_arena_after.json
