#!/bin/bash
# (needs the build of commit b499955: the probe and its knob "tag_reroll" were removed afterwards -- DESIGN 8)
# round 3, forty-third box: does the placement probe of a tag allocation (random byte reads) tell the two modes apart, and does
# drawing twice and keeping the faster one pin the good mode?  Every configuration below re-allocates the tags.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box43
mkdir -p $OUT
cd $R
X="visited=0,lookahead=1,visited_uncached=1"
RG_TRACE_ALLOC=1 timeout 1500 python scripts/exp/k1_ab.py --L 1000 --index-cache /tmp/ix.npz --reps 2 --nbatch 2 \
  --configs "words:visited=0,lookahead=0;a0:visited=0,lookahead=1,tag_reroll=0;x0:$X;a1:visited=0,lookahead=1,tag_reroll=0;x1:$X;a2:visited=0,lookahead=1,tag_reroll=0;x2:$X;a3:visited=0,lookahead=1,tag_reroll=0;x3:$X;a4:visited=0,lookahead=1,tag_reroll=0;x4:$X;r0:visited=0,lookahead=1;x5:$X;r1:visited=0,lookahead=1;x6:$X;r2:visited=0,lookahead=1;x7:$X;r3:visited=0,lookahead=1;x8:$X;r4:visited=0,lookahead=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03_box43/k1_ab.jsonl") if l.startswith('{"config')]
print(" ".join("%s=%.1f"%(r["config"], r["pct_of_8TBs"]) for r in rows if not r["config"].startswith("x")))
PY
grep "probe" $OUT/k1_ab.err | head -30
