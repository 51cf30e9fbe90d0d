#!/bin/bash
# round 4, box 15: the driver's bench command, then the rocprofv3 passes of scripts/profile_r04.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box15
mkdir -p $OUT
cd $R
( time RG_TRACE_ALLOC=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
python scripts/show_bench.py $OUT/bench.json
( time bash scripts/profile_r04.sh ) > $OUT/profile.log 2>&1
tail -5 $OUT/profile.log
