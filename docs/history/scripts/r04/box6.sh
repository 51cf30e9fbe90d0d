#!/bin/bash
# round 4, box 6: buffers balanced over the memory classes (rg_mem.hip) -- A/B against plain allocations, two fresh processes
# each, same box; RG_TRACE_ALLOC shows where the granules came from
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box6
mkdir -p $OUT
cd $R
CFG="default:visited=2;look:visited=0;filter:visited=1"
for rep in 1 2; do
  RG_TRACE_ALLOC=1 timeout 900 python scripts/exp/k1_ab.py --L 50,500,1000,2000 --index-cache /tmp/ix.npz --pipelined --configs "$CFG" \
    > $OUT/balanced_$rep.jsonl 2> $OUT/balanced_$rep.err
  RG_BALANCED_ALLOC=0 RG_TRACE_ALLOC=1 timeout 900 python scripts/exp/k1_ab.py --L 50,500,1000,2000 --index-cache /tmp/ix.npz --pipelined --configs "$CFG" \
    > $OUT/plain_$rep.jsonl 2> $OUT/plain_$rep.err
done
grep "rg_mem" $OUT/balanced_1.err | head -20
tail -2 $OUT/balanced_1.err
for f in balanced_1 plain_1 balanced_2 plain_2; do echo == $f; grep '^{"config' $OUT/$f.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['config'], r['L'], r['pct_of_8TBs'], r['same_ids_hops'], r['same_cmps'])"; done
