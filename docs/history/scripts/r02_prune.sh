#!/bin/bash
# GPU pruning of the build: tests, then the 10M build with and without it (timing on stderr)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/prune; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -m gpu -k "build or pruning" 2>&1 | tail -15 > $o/tests.txt; cat $o/tests.txt
quick="--steps 2 --warmup 1 --sweep=50,500 --no-worstcase --no-fast --no-two-streams --gt-nq 0 --config1-nb 0 --cpu-seconds 0"
RG_BUILD_TIMING=1 timeout 900 python bench.py $quick > $o/gpu_prune.json 2> $o/gpu_prune.err; grep rg_build $o/gpu_prune.err
RG_BUILD_TIMING=1 RG_BUILD_HOST_PRUNE=1 timeout 900 python bench.py $quick > $o/host_prune.json 2> $o/host_prune.err; grep rg_build $o/host_prune.err
python - <<'P'
import json
for n in ("gpu_prune","host_prune"):
    try:
        b=json.loads(open("gpurun_out/prune/%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(b["value"]), b["config"]["setup_seconds"], [(p["L_pq"], round(p["recall_at_10"],4), round(p["mean_evals"])) for p in b["L_pq_sweep"]], b["config"]["workload"][-120:])
    except Exception as e: print(n, "failed", e)
P
