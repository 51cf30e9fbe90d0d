#!/bin/bash
# round 4, box 40: GPU suite + the 56-point sweep with the exact LDS set + tags form in the default rule (L_pq 230 - 290 on the bench index)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box40
mkdir -p $OUT
cd $R
RG_TRACE_ADAPTIVE=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --sweep readme --configs "" --no-worstcase --no-fast --no-two-streams --cpu-seconds 0 --config1-nb 0 --gt-nq 0 > $OUT/bench_sweep_readme.json 2> $OUT/bench.err
python scripts/show_bench.py $OUT/bench_sweep_readme.json | head -62 | awk '{printf "%s ", $0; if (NR%3==0) print ""}' | cut -c1-250
grep "exact LDS set at" $OUT/bench.err | sort | uniq -c | sort -k7 | head -70 | cut -c1-160
