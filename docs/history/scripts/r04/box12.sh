#!/bin/bash
# round 4, box 12: the -m gpu suite from the failing test on, then the driver's bench command, then K2 at small batches
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box12
mkdir -p $OUT
cd $R
( time timeout 3300 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
( time RG_TRACE_ALLOC=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/bench.err
timeout 600 python scripts/exp/gt_small_batch.py 200 10000000 10000,30000,65536,100000 > $OUT/gt_small_batch_200.jsonl 2> $OUT/gt_small.err
timeout 600 python scripts/exp/gt_small_batch.py 512 4000000 10000,65536 l2 > $OUT/gt_small_batch_512.jsonl 2>> $OUT/gt_small.err
cat $OUT/gt_small_batch_200.jsonl $OUT/gt_small_batch_512.jsonl
python scripts/show_bench.py $OUT/bench.json
