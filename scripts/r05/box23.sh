#!/bin/bash
# round 5, box 23: rocprofv3 passes of the rank128 workload (10M x 200, latent rank 128, L_pq 300) and the reference's 56-point sweep at the final code
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
WORKLOADS=rank128 bash scripts/profile_r05.sh 2>&1 | tail -8
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --sweep readme --configs '' --gt-nq 0 --cpu-seconds 0 --no-fast --no-worstcase --no-two-streams --config1-nb 0 --full-out gpurun_out/prof_r05/bench_sweep_readme.json > gpurun_out/prof_r05/bench_sweep_readme_stdout.txt 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/prof_r05/bench_sweep_readme.json'))
print([(p["L_pq"], round(p["pct_of_8000"],1)) for p in d["L_pq_sweep"]])
PY
