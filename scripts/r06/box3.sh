#!/bin/bash
# round 6, box 3: GPU suite at the tree without the 16-row K2 form (d = 512: 384 keys by default), pruning goldens in the caller's form;
# parameters of the 'mixture' family at 1M rows; TCC / SQ counters of the d = 200 random-graph launch; rocprofv3 traces of K2 at d = 512 / 200
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box3
mkdir -p $OUT
cd $R
export RG_FAULT_REPORT=$OUT/fault_report.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest default rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest_gpu.log | tail -6
timeout 900 python scripts/r06/mixture_scan.py 1000000 > $OUT/mixture_scan_1m.jsonl 2> $OUT/mixture_scan.err; echo "mixture scan rc=$?"; cat $OUT/mixture_scan_1m.jsonl; tail -2 $OUT/mixture_scan.err
gt_trace() {  # name d nb nq metric : two launches of one size (scripts/exp/gt_small_batch.py), the kernel's average in the trace
  local name=$1; rm -rf /tmp/rp_gt
  (cd /tmp && TMPDIR=/tmp GT_FORMS=default timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_gt -o s -- python $R/scripts/exp/gt_small_batch.py $2 $3 $4 $5 > $OUT/$name.log 2>&1)
  python scripts/rocprof_summary.py /tmp/rp_gt/*.db > $OUT/$name.txt 2>&1
  grep "rg_gt" $OUT/$name.txt | head -2 | cut -c1-60,98-160; grep frac_of $OUT/$name.log | cut -c1-200
}
gt_trace gt_d512_ip_65536_trace 512 3000000 65536 ip
gt_trace gt_d512_l2_65536_trace 512 3000000 65536 l2
gt_trace gt_d512_ip_10000_trace 512 3000000 10000 ip
gt_trace gt_d512_l2_10000_trace 512 3000000 10000 l2
gt_trace gt_d200_ip_65536_trace 200 10000000 65536 ip
gt_trace gt_d200_ip_10000_trace 200 10000000 10000 ip
WORKLOADS="worst200" PASSES="tcc sq" OUT=$OUT/prof bash scripts/profile_r06.sh 2>&1 | tail -8
grep -h "TCC\|SQ_" $OUT/prof/worst200_tcc.txt $OUT/prof/worst200_sq.txt | grep "rg_search_kernel<false, true, 8" | cut -c1-40,70-200
