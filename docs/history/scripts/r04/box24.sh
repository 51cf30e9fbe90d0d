#!/bin/bash
# round 4, box 24: the driver's bench command at the final code; rocprofv3 passes at the headline, L_pq 200 (exact LDS set that every query
# outgrows) and 500 (exact tags, eight residents)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box24
mkdir -p $OUT
cd $R
( time RG_TRACE_ALLOC=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python scripts/show_bench.py $OUT/bench.json
( time SKIP_GT=1 WORKLOADS="head L200 L500" timeout 2400 bash scripts/profile_r04.sh ) > $OUT/profile.log 2>&1
tail -3 $OUT/profile.log
