#!/bin/bash
# round 6, box 4: the record of the final tree -- GPU suite (default, and with RG_BALANCED_ALLOC=0), smoke, the driver's bench command twice,
# rocprofv3 trace / FETCH / WRITE passes of the headline index (head, L_pq 500 / 1000 / 2000) and of the rank-128 block, K2 traces,
# the mixture family at 10M rows, the random-graph launch in the other visited forms
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box4
mkdir -p $OUT
cd $R
export RG_FAULT_REPORT=$OUT/fault_report.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest default rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest_gpu.log | tail -6
RG_STRESS_ITERS=40 RG_BALANCED_ALLOC=0 timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_balanced_off.log 2>&1; echo "pytest RG_BALANCED_ALLOC=0 rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest_gpu_balanced_off.log | tail -6
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
for i in 1 2; do
  RG_BENCH_PROGRESS=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_default_run$i.json > $OUT/bench_default_run${i}_stdout.txt 2> $OUT/bench_default_run${i}_stderr.txt; echo "bench run $i rc=$?"
  tail -2 $OUT/bench_default_run${i}_stderr.txt; cut -c1-400 $OUT/bench_default_run${i}_stdout.txt
done
gt_trace() {  # name d nb nq metric
  local name=$1; rm -rf /tmp/rp_gt
  (cd /tmp && TMPDIR=/tmp GT_FORMS=default timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_gt -o s -- python $R/scripts/exp/gt_small_batch.py $2 $3 $4 $5 > $OUT/$name.log 2>&1)
  python scripts/rocprof_summary.py /tmp/rp_gt/*.db > $OUT/$name.txt 2>&1
  grep "rg_gt" $OUT/$name.txt | head -3 | cut -c1-60,98-160; grep frac_of $OUT/$name.log | cut -c1-200
}
gt_trace gt_d512_ip_65536_trace 512 3000000 65536 ip
gt_trace gt_d512_l2_65536_trace 512 3000000 65536 l2
gt_trace gt_d512_ip_10000_trace 512 3000000 10000 ip
gt_trace gt_d512_l2_10000_trace 512 3000000 10000 l2
gt_trace gt_d200_ip_65536_trace 200 10000000 65536 ip
gt_trace gt_d200_ip_10000_trace 200 10000000 10000 ip
# the mixture family at the headline's size (one-off: where recall 0.9 lands, what the launches reuse)
timeout 900 python bench.py --in-process --data mixture --rank 128 --steps 10 --warmup 3 --cpu-seconds 4 --gt-nq 0 --no-fast --no-two-streams --config1-nb 0 --configs "" --sweep 10,20,30,50,100,200,300,500,1000 --full-out $OUT/bench_mixture_10m.json > $OUT/bench_mixture_10m_stdout.txt 2> $OUT/bench_mixture_10m_stderr.txt; echo "mixture rc=$?"; cut -c1-300 $OUT/bench_mixture_10m_stdout.txt
# the random-graph launch (frac_hbm_only) in the other forms: exact byte tags with look-ahead, 16 rows in flight
for V in "--visited 0" "--set rows_per_pass=16" "--set rows_per_pass=16,waves_per_cu=12"; do
  timeout 600 python bench.py --in-process --graph random --L 500 --steps 8 --warmup 3 --cpu-seconds 0 --gt-nq 0 --no-fast --no-two-streams --no-worstcase --config1-nb 0 --sweep= --configs= $V --full-out /tmp/w.json > /tmp/w.txt 2>/dev/null
  python - "$V" <<'PY' >> $OUT/worst200_forms.jsonl
import json, sys
d = json.load(open("/tmp/w.json"))
print(json.dumps({"variant": sys.argv[1], "qps": d["value"], "frac": d["roofline"]["frac"], "kernel_ms": d["roofline"]["kernel_ms_avg"], "forms": d["roofline"]["kernel_forms_of_the_batches_so_far"]}))
PY
done
cat $OUT/worst200_forms.jsonl
WORKLOADS="head L500 L1000 L2000 rank128 worst512" PASSES="trace fetch write" OUT=$OUT/prof bash scripts/profile_r06.sh 2>&1 | tail -40
WORKLOADS="worst512" PASSES="sq" OUT=$OUT/prof_sq bash scripts/profile_r06.sh 2>&1 | tail -3
