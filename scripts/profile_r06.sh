#!/bin/bash
# GPU box, round 6: rocprofv3 passes at the round's code (scripts/profile_r05.sh + the workloads round 6 asks about).
#   worst200*     10M x 200 under a RANDOM graph, L_pq 500 (frac_hbm_only at d = 200): the split rows (default), plain 800-B rows
#                 (RG_SPLIT_ROWS=0), rows padded to 1,024 B (--row-stride 256), and the filter + log form instead of the byte tags (--visited 1 ...)
#   mixture       the headline's shape on the low-reuse family, at the L_pq its side block reports
#   head L500 L1000 L2000 webvid laion worst512 rank128: as in round 5
#   passes: trace (--kernel-trace --stats), fetch / write (--pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 on gfx950), tcc, sq
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${OUT:-$R/gpurun_out/prof_r06}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--in-process --steps 8 --warmup 3 --cpu-seconds 0 --gt-nq 0 --no-fast --no-two-streams --no-worstcase --config1-nb 0 --sweep= --configs= ${BENCH_ARGS}"
run() {  # name, command (quoted), rocprof args...
  local name=$1; local cmd=$2; shift; shift
  rm -rf /tmp/rp_$name
  timeout 900 rocprofv3 "$@" -d /tmp/rp_$name -o s -- $cmd --full-out $OUT/$name.bench.json > $OUT/$name.log 2>&1      # (a PMC pass once hung for 47 minutes: every pass has its own limit)
  local db=$(ls /tmp/rp_$name/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocprof_summary.py --json $OUT/$name.pmc.json $db > $OUT/$name.txt 2>&1; fi
  rm -rf /tmp/rp_$name
  grep -v "simple_timer\|SQLite3" $OUT/$name.log | tail -6 > $OUT/$name.log.tail; rm -f $OUT/$name.log
}
for W in ${WORKLOADS:-worst200 worst200_plain worst200_pad1024 worst200_filter}; do
  E=""
  case $W in
    head) A="--index-cache /tmp/bench_ix.npz --L ${L_STAR:-50}";;
    L*) A="--index-cache /tmp/bench_ix.npz --L ${W#L}";;
    webvid) A="--nb 2500000 --dim 512 --metric ip --k 10 --index-cache /tmp/webvid_ix.npz --L 50";;
    laion) A="--nb 2000000 --dim 512 --metric l2 --k 100 --index-cache /tmp/laion_ix.npz --L 150";;
    worst512) A="--nb 2500000 --dim 512 --metric ip --k 10 --graph random --L 500";;
    rank128) A="--rank 128 --index-cache /tmp/rank128_ix.npz --L 300";;
    mixture) A="--data mixture --rank 128 --index-cache /tmp/mixture_ix.npz --L ${L_MIX:-300}";;
    worst200) A="--graph random --L 500";;
    worst200_plain) A="--graph random --L 500"; E="RG_SPLIT_ROWS=0";;
    worst200_pad1024) A="--graph random --L 500 --row-stride 256"; E="RG_SPLIT_ROWS=0";;
    worst200_filter) A="--graph random --L 500 --set adaptive=0,lset=0";;
  esac
  B="env $E python $R/bench.py $COMMON $A"
  # the first command of a workload builds its index into the cache (untimed by rocprof)
  case $W in head|webvid|laion|rank128|mixture) $B --full-out $OUT/${W}_build.bench.json > $OUT/${W}_build.log 2>&1; tail -2 $OUT/${W}_build.log;; esac
  for PASS in ${PASSES:-trace fetch write}; do
    case $PASS in
      trace) run ${W}_trace "$B" --kernel-trace --stats;;
      fetch) run ${W}_fetch "$B" --pmc FETCH_SIZE;;
      write) run ${W}_write "$B" --pmc WRITE_SIZE;;
      sq) run ${W}_sq "$B" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM;;
      tcc) run ${W}_tcc "$B" --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum;;
    esac
  done
  echo "== $W"; grep "rg_search_kernel\|rg_distinct" $OUT/${W}_trace.txt | head -3 | cut -c1-70,98-160
done
python $R/scripts/make_traffic_json.py $(for W in ${WORKLOADS:-worst200 worst200_plain worst200_pad1024 worst200_filter}; do echo $OUT/$W; done) > $OUT/search_traffic.json 2> $OUT/make_traffic.err
python - <<PY
import json
for e in json.load(open("$OUT/search_traffic.json")):
    w=e["workload"]; print(w["nb"],w["dim"],w["graph"],"L",w["L"],"ms %.3f"%(e["kernel_ms_avg_in_the_kernel_trace"] or 0),"alg %.1f GB"%(e["algorithmic_bytes_per_launch"]/1e9),"fetch %.1f write %.1f"%(e["fetch_bytes_corrected"]/1e9,e["write_bytes"]/1e9),"moved/alg %.3f"%e["moved_over_algorithmic"], "moved TB/s %.2f" % ((e["fetch_bytes_corrected"]+e["write_bytes"])/1e9/(e["kernel_ms_avg_in_the_kernel_trace"] or 1e9)))
PY
cat $OUT/make_traffic.err
ls $OUT | wc -l
