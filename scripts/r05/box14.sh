#!/bin/bash
# round 5, box 14: lazy tags with literal waits (tag loads from inline asm) -- parity tests, A/B against the eager form, with 16 and 32 rows in flight
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box14
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 2400 python scripts/exp/k1_ab.py --L 300,500,700,1000,1500,2000 --nbatch 3 --reps 2 --index-cache /tmp/ix.npz \
  --configs "lazy:visited=0,lookahead=1;eager:visited=0,lookahead=5;lazy_r8:visited=0,lookahead=1,rows_per_pass=32;eager_r8:visited=0,lookahead=5,rows_per_pass=32;lazy2:visited=0,lookahead=1;lazy_r8b:visited=0,lookahead=1,rows_per_pass=32" > $OUT/k1_ab_lazy.jsonl 2> $OUT/k1_ab.err
python scripts/r05/ab_table.py $OUT/k1_ab_lazy.jsonl
tail -2 $OUT/k1_ab.err
