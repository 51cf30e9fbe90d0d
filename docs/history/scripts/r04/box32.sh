#!/bin/bash
# round 4, box 32: the driver's sequence at the final code: GPU suite, smoke, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box32
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
RG_TRACE_ALLOC=1 timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python scripts/show_bench.py $OUT/bench.json
