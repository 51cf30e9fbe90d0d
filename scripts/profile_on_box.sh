#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 passes for the search kernel and the GT kernel; only text summaries
# are kept under gpurun_out/ (the rocpd sqlite files are large).  usage: profile_on_box.sh <tag> [search|gt|all]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
WHAT=${2:-all}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args..., -- command
  local name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 "$@" > $OUT/$name.log 2>&1
  local db=$(ls /tmp/rp_$name/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocprof_summary.py --json $OUT/$name.pmc.json $db > $OUT/$name.txt 2>&1; fi
  grep -h '^{' $OUT/$name.log > $OUT/$name.json 2>/dev/null
  rm -rf /tmp/rp_$name
  # keep the logs small
  grep -v "simple_timer\|SQLite3" $OUT/$name.log | tail -20 > $OUT/$name.log.tail; rm -f $OUT/$name.log
}
if [ "$WHAT" = "search" ] || [ "$WHAT" = "all" ]; then
  # the default bench workload (visited mode 2) without the legs that are not the search kernel
  for vis in ${VISITED:-2}; do
    B="python $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --gt-nq 0 --recall-nb 0 --no-other-modes --visited $vis"
    run search_v${vis}_trace --kernel-trace --stats -d /tmp/rp_search_v${vis}_trace -o s -- $B
    run search_v${vis}_fetch --pmc FETCH_SIZE -d /tmp/rp_search_v${vis}_fetch -o s -- $B
    run search_v${vis}_write --pmc WRITE_SIZE -d /tmp/rp_search_v${vis}_write -o s -- $B
    python $R/scripts/make_traffic_json.py $vis $OUT/search_v${vis}_fetch.pmc.json $OUT/search_v${vis}_write.pmc.json > $OUT/search_traffic_v${vis}.json
  done
fi
if [ "$WHAT" = "gt" ] || [ "$WHAT" = "all" ]; then
  G="python $R/scripts/bench_gt.py --nq 65536 --reps 1 --no-warmup"
  run gt_trace --kernel-trace --stats -d /tmp/rp_gt_trace -o s -- $G
  run gt_pmc1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/rp_gt_pmc1 -o s -- $G
  run gt_pmc2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -d /tmp/rp_gt_pmc2 -o s -- $G
  run gt_fetch --pmc FETCH_SIZE -d /tmp/rp_gt_fetch -o s -- $G
fi
ls -la $OUT
