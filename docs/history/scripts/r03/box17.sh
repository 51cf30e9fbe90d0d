#!/bin/bash
# round 3, box 17: the deterministic multi-threaded build on the GPU box (tests, then the bench's 10M build with timing)
cd /root/repo
python -m pytest tests/test_gpu_cli.py -x -q -k "build" 2>&1 | tail -5
RG_BUILD_TIMING=1 python bench.py --gpus 1 --steps 10 --warmup 3 --sweep 50 --cpu-seconds 0 > gpurun_out/bench_det.json 2> gpurun_out/bench_det.err
grep "rg_build" gpurun_out/bench_det.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_det.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['recall_at_10'], d['config']['setup_seconds'])
PY
