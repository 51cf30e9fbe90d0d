#!/bin/bash
# build-thread A/B on the 10M index, then the d = 512 BASELINE shapes with the final kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/more; mkdir -p $o
quick="--steps 2 --warmup 1 --sweep= --L 50 --no-worstcase --no-fast --gt-nq 0 --config1-nb 0 --cpu-seconds 0"
RG_BUILD_TIMING=1 RG_BENCH_BUILD_THREADS=256 timeout 800 python bench.py $quick > $o/build256.json 2> $o/build256.err
grep rg_build $o/build256.err
timeout 900 python bench.py --nb 2500000 --dim 512 --metric ip --no-worstcase --steps 10 --warmup 3 --gt-nq 0 --config1-nb 0 --cpu-seconds 6 > $o/webvid.json 2> $o/webvid.err; echo webvid rc=$?
timeout 1500 python bench.py --nb 10000000 --dim 512 --metric l2 --k 100 --no-worstcase --steps 10 --warmup 3 --gt-nq 0 --config1-nb 0 --cpu-seconds 6 > $o/laion.json 2> $o/laion.err; echo laion rc=$?
python scripts/show_final.py $o/webvid.json 2>&1 | head -3
python scripts/show_final.py $o/laion.json 2>&1 | head -3
