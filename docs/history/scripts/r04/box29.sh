#!/bin/bash
# round 4, box 29: K2 old (rg_gt.hip of the last commit) against new (no scratch reload / vmcnt(0) in the candidate path) in
# alternating processes on ONE box (the MFMA rate differs by 5 % from box to box)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box29
mkdir -p $OUT
cd $R
for rep in 1 2; do
  for lib in old new; do
    L=""; [ $lib = old ] && L=$R/build_ab/librg_hip_oldgt.so
    RG_HIP_LIB=$L GT_FORMS="default:;no_epilogue:RG_GT_DIAG=2" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 8192,10000,65536 2> $OUT/err_$lib$rep.txt | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $OUT/gt_ab.jsonl
  done
done
GT_FORMS="default:;bal_one:RG_GT_BALANCE_ONE=1;bal_one_cand8:RG_GT_BALANCE_ONE=1,RG_GT_CAND=8;cand8:RG_GT_CAND=8" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 10000,30000 2> $OUT/err_bal.txt | tee $OUT/gt_bal.jsonl
GT_FORMS="default:" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 10000,65536 l2 2> $OUT/err_l2.txt | sed "s/^{/{\"lib\": \"new\", \"metric\": \"l2\", /" | tee $OUT/gt_l2.jsonl
RG_HIP_LIB=$R/build_ab/librg_hip_oldgt.so GT_FORMS="default:" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 10000,65536 l2 2> $OUT/err_l2o.txt | sed "s/^{/{\"lib\": \"old\", \"metric\": \"l2\", /" | tee -a $OUT/gt_l2.jsonl
