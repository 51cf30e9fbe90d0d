#!/bin/bash
# round 6, box 8: the driver's sequence at the final commit -- GPU suite with -x, smoke, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box8
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_default.json > $OUT/bench_default_stdout.txt 2> $OUT/bench_default_stderr.txt; echo "bench rc=$?"; tail -2 $OUT/bench_default_stderr.txt; cat $OUT/bench_default_stdout.txt | cut -c1-6000
