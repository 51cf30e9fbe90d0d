#!/bin/bash
# round 4, box 23: the allocator with one granule-by-granule walk per pool epoch -- stress, then three fresh processes of the bench after an
# in-process build (classes per buffer, wide beams), then the search suites
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box23
mkdir -p $OUT
cd $R
for i in 1 2; do timeout 600 python scripts/exp/mem_stress.py 6 > $OUT/stress_$i.log 2> $OUT/stress_$i.err; echo "stress $i rc=$?"; tail -1 $OUT/stress_$i.log; grep -v '"mismatching_words": 0' $OUT/stress_$i.log | grep round | tail -2; done
for i in 1 2 3; do
  ( time RG_TRACE_ALLOC=1 timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --sweep 500,1000,2000 --L 50 --no-worstcase --no-fast --no-two-streams --gt-nq 0 --cpu-seconds 0 --config1-nb 0 --configs '' ) > $OUT/bench_run$i.json 2> $OUT/bench_run$i.err
  python -c "
import json
l=[x for x in open('$OUT/bench_run$i.json') if x.startswith('{')]
r=json.loads(l[-1]); print('run $i', round(r['roofline']['frac'],4), [(p['L_pq'], round(p['pct_of_8000'],1)) for p in r['L_pq_sweep']], r['device_memory'], r['config']['setup_seconds'])"
  grep "rg_mem\] [0-9d]" $OUT/bench_run$i.err | cut -c1-150; grep real $OUT/bench_run$i.err
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_concurrency.py tests/test_gpu_baseline_shapes.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -n "passed\|failed" $OUT/pytest.log | tail -2
