#!/bin/bash
# round 4, box 30: K2 old (rg_gt.hip of commit 2b21d04) against new (filter verdict as an OR of ballots, two-instruction maxima) in
# alternating processes on ONE box (the MFMA rate differs by 5 % from box to box)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box30
mkdir -p $OUT
cd $R
for rep in 1 2; do
  for lib in old new; do
    L=""; [ $lib = old ] && L=$R/build_ab/librg_hip_oldgt.so
    RG_HIP_LIB=$L GT_FORMS="default:;cand8:RG_GT_CAND=8" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 8192,10000,65536 2> $OUT/err_$lib$rep.txt | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $OUT/gt_ab.jsonl
  done
done
GT_FORMS="default:" timeout 300 python scripts/exp/gt_small_batch.py 512 4000000 10000,65536 l2 2> $OUT/err_512n.txt | sed "s/^{/{\"lib\": \"new\", \"metric\": \"l2\", /" | tee $OUT/gt_d512_l2.jsonl
RG_HIP_LIB=$R/build_ab/librg_hip_oldgt.so GT_FORMS="default:" timeout 300 python scripts/exp/gt_small_batch.py 512 4000000 10000,65536 l2 2> $OUT/err_512o.txt | sed "s/^{/{\"lib\": \"old\", \"metric\": \"l2\", /" | tee -a $OUT/gt_d512_l2.jsonl
timeout 600 python -m pytest tests/test_gpu_groundtruth.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
