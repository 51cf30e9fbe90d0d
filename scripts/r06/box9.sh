#!/bin/bash
# round 6, box 9: soak of the final commit -- the GPU suite three times, the bench once
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box9
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  RG_FAULT_REPORT=$OUT/fault_report_$i.txt timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_$i.log 2>&1; echo "pytest run $i rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest_gpu_$i.log | tail -2
done
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_default.json > $OUT/bench_default_stdout.txt 2> $OUT/bench_default_stderr.txt; echo "bench rc=$?"; cut -c1-200 $OUT/bench_default_stdout.txt
