#!/bin/bash
# round 5, box 3: where a hop of the look-ahead form spends its cycles at wide beams (instrumented build), current code
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box3
mkdir -p $OUT
cd $R
RG_HIP_LIB=$R/roargraph_amd/librg_hip_prof.so timeout 1200 python scripts/exp/k1_phases.py --nb 10000000 --Ls 300,500,1000,2000 --modes 0 --set lookahead=1 --save /tmp/ix --out $OUT/phases_look_hub.json > $OUT/phases_look_hub.log 2>&1
RG_HIP_LIB=$R/roargraph_amd/librg_hip_prof.so timeout 600 python scripts/exp/k1_phases.py --nb 10000000 --Ls 500,2000 --modes 0 --set lookahead=1,hub_bits=0 --load /tmp/ix --out $OUT/phases_look_nohub.json > $OUT/phases_look_nohub.log 2>&1
python scripts/exp/show_phases.py $OUT/phases_look_hub.json $OUT/phases_look_nohub.json
tail -3 $OUT/phases_look_hub.log
