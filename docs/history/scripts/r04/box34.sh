#!/bin/bash
# round 4, box 34: the reference's own 56-point L_pq list (README.md:118) on the bench index at the final code
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box34
mkdir -p $OUT
cd $R
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --sweep readme --configs "" --no-worstcase --no-fast --no-two-streams --cpu-seconds 0 --config1-nb 0 --gt-nq 0 > $OUT/bench_sweep_readme.json 2> $OUT/bench.err
python scripts/show_bench.py $OUT/bench_sweep_readme.json | head -70
