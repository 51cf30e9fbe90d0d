#!/bin/bash
# rows-in-flight sweep of K1 on small-degree random graphs (latency-bound regime) and the default graph
R=${GRAFT_REPO_ROOT:-/root/repo}
for deg in 16 40; do
  for L in 100 500; do
    for rpp in 4 8 16; do
      python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --gt-nq 0 --recall-nb 0 --no-other-modes --deg $deg --L $L --rows-per-pass $rpp 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); c=d['config']
print('deg $deg L $L rows/pass $rpp: %.0f QPS  %.2f ms  evals %.0f hops %.0f  %.0f GB/s' % (d['value'], d['ms_per_step'], c['mean_evals_per_query'], c['mean_hops'], d['roofline']['achieved']))"
    done
  done
done
