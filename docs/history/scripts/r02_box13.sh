#!/bin/bash
# 10M genuine index: which visited form wins at L_pq 300-1000, and ring depth at 500
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_box13
mkdir -p $OUT
cd $R
python scripts/exp/k1_phases.py --nb 10000000 --save /tmp/ix10 --Ls 300,500,700,1000,2000 --modes 0,1,2 --out $OUT/base.json > $OUT/base.log 2>&1
for S in rows_per_pass=8 rows_per_pass=32 exact_filter=0; do
python scripts/exp/k1_phases.py --nb 10000000 --load /tmp/ix10 --Ls 300,500,1000 --modes 0,1 --set $S --out $OUT/$S.json > $OUT/$S.log 2>&1
done
RG_HIP_LIB=$R/roargraph_amd/librg_hip_prof.so python scripts/exp/k1_phases.py --nb 10000000 --load /tmp/ix10 --Ls 500,1000 --modes 0,1 --out $OUT/prof.json > $OUT/prof.log 2>&1
