#!/bin/bash
# round 4, box 45: rocprofv3 kernel traces of K2 at the final code, one query count per run
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box45
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for NQ in 10000 65536; do
  rm -rf /tmp/rp_gt_$NQ
  GT_FORMS=default timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_gt_$NQ -o s -- python $R/scripts/exp/gt_small_batch.py 200 10000000 $NQ > $OUT/gt_$NQ.log 2>&1
  db=$(ls /tmp/rp_gt_$NQ/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db > $OUT/gt_${NQ}_trace.txt 2>&1
  head -5 $OUT/gt_${NQ}_trace.txt | cut -c1-170; grep '^{' $OUT/gt_$NQ.log
  rm -f $OUT/gt_$NQ.log
done
