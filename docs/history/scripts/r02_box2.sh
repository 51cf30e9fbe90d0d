#!/bin/bash
# knob sensitivity of K1 on the genuine 2M index (instrumented build): ring depth and resident waves at large L_pq
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_box2
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
export RG_HIP_LIB=$R/roargraph_amd/librg_hip_prof.so
python scripts/exp/k1_phases.py --nb 2000000 --save /tmp/ix --Ls 500,1000,2000 --modes 0 --out $OUT/base.json > $OUT/base.log 2>&1
for S in rows_per_pass=16 rows_per_pass=4 rows_per_pass=16,waves_per_cu=4 waves_per_cu=4 waves_per_cu=3 exact_filter=0; do
  python scripts/exp/k1_phases.py --nb 2000000 --load /tmp/ix --Ls 500,1000,2000 --modes 0 --set $S --out $OUT/$S.json > $OUT/$S.log 2>&1
done
python scripts/exp/k1_phases.py --nb 2000000 --load /tmp/ix --Ls 500,1000,2000 --modes 1 --set rows_per_pass=16 --out $OUT/m1_rpp16.json > $OUT/m1_rpp16.log 2>&1
