#!/bin/bash
# K1 with the speculative second expansion: parity suite, then phase profile and product QPS with the knob off / on
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG:-r02_box3}
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
NB=${NB:-2000000}
LS=${LS:-100,500,1000,2000}
python scripts/exp/k1_phases.py --nb $NB --save /tmp/ix --Ls 100 --modes 1 > $OUT/build.log 2>&1
for S in ${SETS:-waves_per_cu=0}; do
  RG_HIP_LIB=$R/roargraph_amd/librg_hip_prof.so python scripts/exp/k1_phases.py --nb $NB --load /tmp/ix --Ls $LS --modes ${MODES:-0,1} --set $S --out $OUT/prof_$S.json > $OUT/prof_$S.log 2>&1
  python scripts/exp/k1_phases.py --nb $NB --load /tmp/ix --Ls $LS --modes ${MODES:-0,1,2} --set $S --out $OUT/prod_$S.json > $OUT/prod_$S.log 2>&1
done
