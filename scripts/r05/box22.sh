#!/bin/bash
# round 5, box 22: K2 kernel traces at 10,000 and 65,536 queries (final code), then the driver's bench command twice (stability)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box22
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for NQ in 10000 65536; do
  rm -rf /tmp/rp_gt
  GT_FORMS=default timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_gt -o s -- python $R/scripts/exp/gt_small_batch.py 200 10000000 $NQ > $OUT/gt_${NQ}.log 2>&1
  db=$(ls /tmp/rp_gt/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db > $OUT/gt_${NQ}_trace.txt 2>&1
  grep '^{' $OUT/gt_${NQ}.log; grep "rg_gt_rs_kernel" $OUT/gt_${NQ}_trace.txt | head -2
done
cd $R
for i in 1 2; do
  RG_BENCH_PROGRESS=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench${i}_full.json > $OUT/bench${i}_stdout.txt 2> $OUT/bench${i}_stderr.txt
  echo "bench run $i rc=$? bytes=$(wc -c < $OUT/bench${i}_stdout.txt)"; grep -v "^\[bench" $OUT/bench${i}_stderr.txt | tail -5
done
