#!/bin/bash
# round 4, box 17: (a) exact-tag form at L_pq 300 - 700: eight residents with 32 rows in flight and a larger bit screen against the default;
# (b) K2 trace per query count (one form, one shape per pass); (c) three fresh processes of the bench at the wide beams (placement pinned?)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box17
mkdir -p $OUT
cd $R
timeout 900 python scripts/exp/k1_ab.py --L 300,500,700 --index-cache /tmp/ix.npz --pipelined --nbatch 4 \
  --configs "look:visited=0;look_r8:visited=0,rows_per_pass=32;look_r8_w8:visited=0,rows_per_pass=32,waves_per_cu=8;look_r4_w12:visited=0,rows_per_pass=16,waves_per_cu=12;look_r4_w8:visited=0,rows_per_pass=16,waves_per_cu=8;default:visited=2;filter:visited=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
Ls=sorted({r['L'] for r in rows}); cf=[]
for r in rows:
    if r['config'] not in cf: cf.append(r['config'])
print('%-14s'%'cfg'+''.join('%8d'%L for L in Ls))
for c in cf: print('%-14s'%c+''.join('%8.1f'%next((r['pct_of_8TBs'] for r in rows if r['config']==c and r['L']==L),0) for L in Ls))
print('exact', all(r['same_ids_hops'] in (None,True) for r in rows))"
cd /tmp && export TMPDIR=/tmp
for nq in 10000 65536; do
  rm -rf /tmp/rp_gt_$nq
  GT_FORMS=default timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_gt_$nq -o s -- python $R/scripts/exp/gt_small_batch.py 200 10000000 $nq > $OUT/gt_${nq}.log 2>&1
  db=$(ls /tmp/rp_gt_$nq/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocprof_summary.py $db > $OUT/gt_${nq}_trace.txt 2>&1; fi
  grep '^{' $OUT/gt_${nq}.log > $OUT/gt_${nq}.jsonl; rm -f $OUT/gt_${nq}.log
  head -4 $OUT/gt_${nq}_trace.txt | cut -c1-160
done
cd $R
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --sweep 500,1000,2000 --L 50 --index-cache /tmp/bench_ix.npz --no-worstcase --no-fast --no-two-streams --gt-nq 0 --cpu-seconds 0 --config1-nb 0 --configs '' > $OUT/bench_run$i.json 2> $OUT/bench_run$i.err
  python -c "
import json
l=[x for x in open('$OUT/bench_run$i.json') if x.startswith('{')]
r=json.loads(l[-1]); print('run $i', round(r['roofline']['frac'],4), [(p['L_pq'], round(p['pct_of_8000'],1)) for p in r['L_pq_sweep']], r['device_memory'])"
done
