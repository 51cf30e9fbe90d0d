#!/bin/bash
# round 3, thirty-ninth box: the same launch over several ALLOCATIONS of the 20 GB of byte tags in one process (toggling
# visited_uncached frees the buffers; RG_TRACE_ALLOC=1 prints where each allocation landed)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box39
mkdir -p $OUT
cd $R
RG_TRACE_ALLOC=1 timeout 1500 python scripts/exp/k1_ab.py --L 1000,2000 --index-cache /tmp/ix.npz --reps 2 --nbatch 2 \
  --configs "words:visited=0,lookahead=0;l0:visited=0,lookahead=1;x1:visited=0,lookahead=1,visited_uncached=1;l1:visited=0,lookahead=1;x2:visited=0,lookahead=1,visited_uncached=1;l2:visited=0,lookahead=1;x3:visited=0,lookahead=1,visited_uncached=1;l3:visited=0,lookahead=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box39/k1_ab.jsonl") if l.startswith('{"config')]
for r in rows: print(r["config"], r["L"], r["ms"], r["pct_of_8TBs"])
PY
grep "visited" $OUT/k1_ab.err | head -40
