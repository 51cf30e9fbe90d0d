#!/bin/bash
# round 4, box 2: map of the device memory -- speed of K1's access mix by WHERE rows and tags were allocated
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box2
mkdir -p $OUT
cd $R
timeout 900 scripts/exp/bin/alloc_map 16 17 > $OUT/alloc_map_16.jsonl 2> $OUT/alloc_map_16.err
timeout 900 scripts/exp/bin/alloc_map 8 34 > $OUT/alloc_map_8.jsonl 2> $OUT/alloc_map_8.err
cat $OUT/alloc_map_16.jsonl; cat $OUT/alloc_map_8.jsonl | tail -45
