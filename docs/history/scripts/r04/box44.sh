#!/bin/bash
# round 4, box 44: last check of the tree as committed: GPU suite + smoke
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box44
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
