#!/bin/bash
# round 4, box 33: rocprofv3 summaries at the final code: headline (trace, fetch, write), L_pq 100 / 200 (trace, fetch), K2 traces
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
WORKLOADS="head L100 L200" PASSES="trace fetch write" SKIP_GT_SQ=1 SKIP_CALIB=1 timeout 2400 bash scripts/profile_r04.sh > $R/gpurun_out/prof_r04_box33.log 2>&1
tail -5 $R/gpurun_out/prof_r04_box33.log
cat $R/gpurun_out/prof_r04/search_traffic.json | head -60
