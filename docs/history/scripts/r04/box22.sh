#!/bin/bash
# round 4, box 22: parity suites of the search path, then the driver's bench command (new rule of the exact LDS set; longer allocator walks)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box22
mkdir -p $OUT
cd $R
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
grep -n "passed\|failed" $OUT/pytest.log | tail -3
( time RG_TRACE_ALLOC=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python scripts/show_bench.py $OUT/bench.json
grep "rg_mem\] [0-9d]" $OUT/bench.err | head -8 | cut -c1-150
