#!/bin/bash
# round 5, box 5: the whole -m gpu suite + smoke + the driver's bench command (compact line on stdout, full record in a file)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box5
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time RG_BENCH_PROGRESS=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_full.json ) > $OUT/bench_stdout.txt 2> $OUT/bench_stderr.txt
wc -c $OUT/bench_stdout.txt; tail -c 4500 $OUT/bench_stdout.txt; grep -v "^\[bench" $OUT/bench_stderr.txt | tail -12
python scripts/show_bench.py $OUT/bench_full.json 2>&1 | tail -40
