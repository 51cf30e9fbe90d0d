#!/bin/bash
# the d = 512 BASELINE shapes (config 5: webvid-2.5M IP, config 4: laion-10M L2 top-100) with the final kernels and build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/shapes; mkdir -p $o
RG_BUILD_TIMING=1 timeout 900 python bench.py --nb 2500000 --dim 512 --metric ip --no-worstcase --steps 10 --warmup 3 --gt-nq 0 --config1-nb 0 --cpu-seconds 6 > $o/webvid.json 2> $o/webvid.err; echo webvid rc=$?
RG_BUILD_TIMING=1 timeout 1500 python bench.py --nb 10000000 --dim 512 --metric l2 --k 100 --no-worstcase --steps 10 --warmup 3 --gt-nq 0 --config1-nb 0 --cpu-seconds 6 > $o/laion.json 2> $o/laion.err; echo laion rc=$?
for n in webvid laion; do python scripts/show_final.py $o/$n.json 2>&1 | head -3 | cut -c1-900; grep "phase 3\|phase 1 " $o/$n.err | tail -3; done
