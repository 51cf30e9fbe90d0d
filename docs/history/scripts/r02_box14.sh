#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_box14
mkdir -p $OUT
cd $R
python scripts/exp/k1_phases.py --nb 2000000 --save /tmp/ix --Ls 30,50,100,200,300 --modes 2,1 --out $OUT/base.json > $OUT/base.log 2>&1
for S in rows_per_pass=8 rows_per_pass=4 rows_per_pass=8,waves_per_cu=24 rows_per_pass=16,waves_per_cu=12; do
python scripts/exp/k1_phases.py --nb 2000000 --load /tmp/ix --Ls 30,50,100,200,300 --modes 2,1 --set $S --out $OUT/$S.json > $OUT/$S.log 2>&1
done
