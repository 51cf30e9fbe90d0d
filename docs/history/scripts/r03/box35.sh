#!/bin/bash
# round 3, thirty-fifth box: the d = 512 BASELINE shapes (config 5: webvid-2.5M IP; config 4: laion-10M L2 top-100) with the
# round's final kernels (streamed gather, byte-tag look-ahead form, filter fill)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box35
mkdir -p $OUT
cd $R
RG_BUILD_TIMING=1 timeout 900 python bench.py --nb 2500000 --dim 512 --metric ip --no-worstcase --steps 10 --warmup 3 --gt-nq 0 --config1-nb 0 --cpu-seconds 6 > $OUT/bench_webvid_shape.json 2> $OUT/webvid.err; echo webvid rc=$?
RG_BUILD_TIMING=1 timeout 1500 python bench.py --nb 10000000 --dim 512 --metric l2 --k 100 --no-worstcase --steps 10 --warmup 3 --gt-nq 0 --config1-nb 0 --cpu-seconds 6 > $OUT/bench_laion_shape.json 2> $OUT/laion.err; echo laion rc=$?
python - <<'PY'
import json
for n in ("bench_webvid_shape","bench_laion_shape"):
    try:
        d=json.loads(open("gpurun_out/r03_box35/%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"]), d["config"].get("L_pq"), d["config"].get("recall_at_10"), round(d["roofline"]["frac"],4))
        for p in d["L_pq_sweep"]: print("  ", p["L_pq"], round(p["qps"]), round(p["recall_at_10"],4), round(p["pct_of_8000"],1))
    except Exception as e: print(n, "no line:", e)
PY
tail -q -n 2 $OUT/webvid.err $OUT/laion.err
