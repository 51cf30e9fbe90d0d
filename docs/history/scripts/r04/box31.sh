#!/bin/bash
# round 4, box 31: where K2's epilogue time goes -- the instrumented instantiation (RG_GT_PROF=1: s_memtime sums per workgroup)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box31
mkdir -p $OUT
cd $R
for K in 100 10; do
  GT_K=$K GT_FORMS="default:;prof:RG_GT_PROF=1" timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 8192,10000,65536 2> $OUT/prof_K$K.txt | tee $OUT/gt_prof_K$K.jsonl
  grep "rg_gt prof" $OUT/prof_K$K.txt
done
