#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_box15
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
python scripts/exp/k1_phases.py --nb 2000000 --save /tmp/ix --Ls 30,50,100,200,300,500 --modes 2,1 --out $OUT/base.json > $OUT/base.log 2>&1
