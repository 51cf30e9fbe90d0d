#!/bin/bash
# round 4, box 25: the -m gpu suite, smoke() and a short bench at the last code of the round (tail-count code removed, pool cap)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box25
mkdir -p $OUT
cd $R
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
grep -n "passed\|failed" $OUT/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( time RG_TRACE_ALLOC=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-fast --config1-nb 0 --configs '' ) > $OUT/bench.json 2> $OUT/bench.err
python scripts/show_bench.py $OUT/bench.json
