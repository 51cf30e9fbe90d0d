#!/bin/bash
# round 3, tenth box: evidence runs -- hub-first layout A/B, latent-rank sensitivity of the headline, the d = 512 shapes, the
# README workflow over files with the CLI twins
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box10
mkdir -p $OUT
cd $R
timeout 900 python scripts/exp/hubfirst_ab.py --L 50,200,500 > $OUT/hubfirst_ab.jsonl 2> $OUT/hubfirst_ab.err; echo "hubfirst rc=$?"; cat $OUT/hubfirst_ab.jsonl
for RK in 64 128; do
  timeout 900 python bench.py --rank $RK --steps 10 --warmup 3 --no-worstcase --no-fast --no-two-streams --cpu-seconds 0 --gt-nq 0 --config1-nb 0 \
     --sweep 20,50,100,200,500,1000,2000 > $OUT/bench_rank$RK.json 2> $OUT/bench_rank$RK.err; echo "rank $RK rc=$?"
done
RG_BUILD_TIMING=1 timeout 900 python bench.py --nb 2500000 --dim 512 --metric ip --no-worstcase --steps 10 --warmup 3 --gt-nq 0 --config1-nb 0 --cpu-seconds 6 > $OUT/bench_webvid_shape.json 2> $OUT/webvid.err; echo webvid rc=$?
RG_BUILD_TIMING=1 timeout 1500 python bench.py --nb 10000000 --dim 512 --metric l2 --k 100 --no-worstcase --steps 10 --warmup 3 --gt-nq 0 --config1-nb 0 --cpu-seconds 6 > $OUT/bench_laion_shape.json 2> $OUT/laion.err; echo laion rc=$?
python - <<'PY'
import json
for n in ("bench_rank64","bench_rank128","bench_webvid_shape","bench_laion_shape"):
    try:
        d=json.loads(open("gpurun_out/r03_box10/%s.json"%n).read().strip().splitlines()[-1])
        r=d["roofline"]; ru=r.get("reuse") or {}
        print(n, round(d["value"]), d["config"]["L_pq"], round(d["config"]["recall_at_10"],4), "frac %.3f"%r["frac"], "distinct %.3f"%(r.get("distinct_rows_frac") or 0), ru.get("share_of_reads_to_top_rows"), d["config"]["workload"][-120:])
        print("   ", [(p["L_pq"], round(p["recall_at_10"],3), round(p["pct_of_8000"],1)) for p in d["L_pq_sweep"]])
        if d.get("cpu_baseline"): print("    cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("value_without_prefetch"))
    except Exception as e: print(n, "failed:", e)
PY
O=gpurun_out/e2e_cli bash scripts/e2e_cli_t2i10m.sh > $OUT/e2e_cli.txt 2>&1; tail -32 $OUT/e2e_cli.txt
