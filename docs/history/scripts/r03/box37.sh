#!/bin/bash
# round 3, thirty-seventh box: hub-first numbering under the BYTE-tag exact set (20 GB of tags indexed by id): how much of a
# wide beam's time is the placement of the hot nodes' tags (pages, TLB reach)?  Not a product layout (tie order) -- a probe.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box37
mkdir -p $OUT
cd $R
timeout 1500 python scripts/exp/hubfirst_ab.py --L 500,1000,2000 --nbatch 3 --reps 2 --knobs visited=0,lookahead=1 --index-cache /tmp/ix.npz > $OUT/hubfirst_bytes.jsonl 2> $OUT/hubfirst_bytes.err
cat $OUT/hubfirst_bytes.jsonl | cut -c1-220
tail -2 $OUT/hubfirst_bytes.err
