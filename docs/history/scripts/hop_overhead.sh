#!/bin/bash
# K1 in the instruction-bound regime: random graphs of small out-degree (few fresh neighbours per hop).
# usage (on the GPU box): bash scripts/hop_overhead.sh [pmc]
R=${GRAFT_REPO_ROOT:-/root/repo}
for deg in 8 16 24; do
  for L in 100 500; do
    python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --gt-nq 0 --recall-nb 0 --no-other-modes --deg $deg --L $L 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); c=d['config']
print('deg $deg L $L: %.0f QPS  %.2f ms  evals %.0f hops %.0f  -> %.2f us/hop/wave-slot, %.0f GB/s' % (d['value'], d['ms_per_step'], c['mean_evals_per_query'], c['mean_hops'], d['ms_per_step']*1e3*4096/(10000*c['mean_hops']), d['roofline']['achieved']))"
  done
done
if [ "$1" = "pmc" ]; then
  cd /tmp && export TMPDIR=/tmp
  B="python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --gt-nq 0 --recall-nb 0 --no-other-modes --deg 16 --L 500"
  rm -rf /tmp/rp_h; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/rp_h -o s -- $B > /dev/null 2>&1
  python $R/scripts/rocprof_summary.py $(ls /tmp/rp_h/*.db | head -1) | grep "rg_search_kernel" | cut -c1-150
fi
