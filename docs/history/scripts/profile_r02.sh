#!/bin/bash
# GPU box, round 2: rocprofv3 passes of the DEFAULT bench workload at its headline beam width (genuine 10M index; the
# first pass builds it, the others load it from --index-cache) and of the worst-case block (random graph, L_pq = 500);
# text summaries kept under gpurun_out/prof_r02 for profiles/r02/.  L_STAR = the headline beam width of the full run.
#   trace        --kernel-trace --stats                 (average kernel durations)
#   fetch/write  --pmc FETCH_SIZE / WRITE_SIZE          (HBM traffic; FETCH_SIZE x2 on gfx950)
#   sq           --pmc SQ_* issue/wait counters          (where the wave cycles go)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 5 --warmup 2 --cpu-seconds 0 --gt-nq 0 --no-fast --no-two-streams --no-worstcase --config1-nb 0 --sweep= ${BENCH_ARGS}"
run() {  # name, bench args (quoted), rocprof args...
  local name=$1; local bargs=$2; shift; shift
  rm -rf /tmp/rp_$name
  rocprofv3 "$@" -d /tmp/rp_$name -o s -- python $R/bench.py $COMMON $bargs > $OUT/$name.log 2>&1
  local db=$(ls /tmp/rp_$name/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocprof_summary.py --json $OUT/$name.pmc.json $db > $OUT/$name.txt 2>&1; fi
  grep -h '^{' $OUT/$name.log > $OUT/$name.bench.json 2>/dev/null
  rm -rf /tmp/rp_$name
  grep -v "simple_timer\|SQLite3" $OUT/$name.log | tail -12 > $OUT/$name.log.tail; rm -f $OUT/$name.log
}
# the index is built once, outside the profiler (its phase-3 searches run K1 in build mode and would pollute the kernel stats)
python $R/bench.py $COMMON --index-cache /tmp/bench_ix.npz --L ${L_STAR:-50} > $OUT/build_run.log 2>&1
for W in head:"--index-cache /tmp/bench_ix.npz --L ${L_STAR:-50}" L500:"--index-cache /tmp/bench_ix.npz --L 500" L1000:"--index-cache /tmp/bench_ix.npz --L 1000" L2000:"--index-cache /tmp/bench_ix.npz --L 2000" worst:"--graph random --L 500"; do
  N=${W%%:*}; A=${W#*:}
  run ${N}_trace "$A" --kernel-trace --stats
  run ${N}_fetch "$A" --pmc FETCH_SIZE
  run ${N}_write "$A" --pmc WRITE_SIZE
  run ${N}_sq "$A" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
done
python $R/scripts/make_traffic_json.py $OUT/head $OUT/L500 $OUT/L1000 $OUT/L2000 $OUT/worst > $OUT/search_traffic.json 2> $OUT/make_traffic.err
ls -la $OUT
