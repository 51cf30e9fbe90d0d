#!/bin/bash
# GPU box, round 4: rocprofv3 passes of the bench workload (genuine 10M index; built once into --index-cache) at the headline
# beam width, L_pq = 500 / 1000 / 2000 and the worst case; text summaries under gpurun_out/prof_r04 for profiles/r04/.
#   trace        --kernel-trace --stats                 (average kernel durations)
#   fetch/write  --pmc FETCH_SIZE / WRITE_SIZE          (fabric traffic; FETCH_SIZE x2 on gfx950, calibrated below)
#   sq           --pmc SQ_* issue/wait counters          (where the wave cycles go)
#   tcc          --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum   (L2 hit rate; read requests to the fabric)
#   calib_*      scripts/exp/calib_fetch.py under FETCH_SIZE and under the raw TCC request counters (known byte count)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 8 --warmup 3 --cpu-seconds 0 --gt-nq 0 --no-fast --no-two-streams --no-worstcase --config1-nb 0 --sweep= --configs= ${BENCH_ARGS}"
run() {  # name, command (quoted), rocprof args...
  local name=$1; local cmd=$2; shift; shift
  rm -rf /tmp/rp_$name
  timeout 900 rocprofv3 "$@" -d /tmp/rp_$name -o s -- $cmd > $OUT/$name.log 2>&1      # (a PMC pass once hung for 47 minutes: every pass has its own limit)
  local db=$(ls /tmp/rp_$name/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocprof_summary.py --json $OUT/$name.pmc.json $db > $OUT/$name.txt 2>&1; fi
  grep -h '^{' $OUT/$name.log > $OUT/$name.bench.json 2>/dev/null
  rm -rf /tmp/rp_$name
  grep -v "simple_timer\|SQLite3" $OUT/$name.log | tail -12 > $OUT/$name.log.tail; rm -f $OUT/$name.log
}
python $R/bench.py $COMMON --index-cache /tmp/bench_ix.npz --L ${L_STAR:-50} > $OUT/build_run.log 2>&1
for W in ${WORKLOADS:-head L500 L1000 L2000 worst}; do
  case $W in
    head) A="--index-cache /tmp/bench_ix.npz --L ${L_STAR:-50}";;
    L*) A="--index-cache /tmp/bench_ix.npz --L ${W#L}";;
    worst) A="--graph random --L 500";;
  esac
  B="python $R/bench.py $COMMON $A"
  for PASS in ${PASSES:-trace fetch write sq tcc}; do
    case $PASS in
      trace) run ${W}_trace "$B" --kernel-trace --stats;;
      fetch) run ${W}_fetch "$B" --pmc FETCH_SIZE;;
      write) run ${W}_write "$B" --pmc WRITE_SIZE;;
      sq) run ${W}_sq "$B" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM;;
      tcc) run ${W}_tcc "$B" --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum;;
    esac
  done
done
if [ -z "$SKIP_GT" ]; then
# K2 at the two query counts VERDICT r3 #6 names (10,000 = the eval-side truth and every tail batch; 65,536 = a streamed batch)
export GT_FORMS=default
run gt_trace "python $R/scripts/exp/gt_small_batch.py 200 10000000 10000,65536" --kernel-trace --stats
[ -z "$SKIP_GT_SQ" ] && run gt_sq "python $R/scripts/exp/gt_small_batch.py 200 10000000 10000,65536" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
[ -z "$SKIP_CALIB" ] && run calib_fetch "python $R/scripts/exp/calib_fetch.py" --pmc FETCH_SIZE
[ -z "$SKIP_CALIB" ] && run calib_tcc "python $R/scripts/exp/calib_fetch.py" --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_HIT_sum
fi
python $R/scripts/make_traffic_json.py $(for W in ${WORKLOADS:-head L500 L1000 L2000 worst}; do echo $OUT/$W; done) > $OUT/search_traffic.json 2> $OUT/make_traffic.err
ls -la $OUT
