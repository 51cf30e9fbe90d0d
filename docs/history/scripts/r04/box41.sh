#!/bin/bash
# round 4, box 41: rocprofv3 summaries at the final code for the widths the combined forms serve: L_pq 300 (set in front of the look-ahead tags), 500
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
WORKLOADS="L300 L500" PASSES="trace fetch write" SKIP_GT=1 timeout 2000 bash scripts/profile_r04.sh > $R/gpurun_out/prof_r04_box41.log 2>&1
tail -3 $R/gpurun_out/prof_r04_box41.log
for W in L300 L500; do grep "rg_search_kernel" $R/gpurun_out/prof_r04/${W}_trace.txt | head -4 | cut -c1-170; done
python - <<'PY'
import json
for e in json.load(open('gpurun_out/prof_r04/search_traffic.json')):
    print(e['workload']['L'], e['kernels'], round(e['kernel_ms_avg_under_rocprof'],3), round(e['fetch_bytes_corrected']/1e9,2), round(e['write_bytes']/1e9,3), round(e['algorithmic_bytes_per_launch']/1e9,2))
PY
