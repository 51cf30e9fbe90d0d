#!/bin/bash
# round 4, box 28: (a) K2 with the candidate store as a global store from an address made in place (no scratch reload, no
# vmcnt(0), no flat store); (b) K1 after the counters of the exact-LDS-set form left scratch memory (one scratch load + vmcnt(0) +
# store per hop before): the whole GPU suite, then the bench without its side blocks
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box28
mkdir -p $OUT
cd $R
GT_FORMS="default:;cand8:RG_GT_CAND=8;no_epilogue:RG_GT_DIAG=2" \
  timeout 600 python scripts/exp/gt_small_batch.py 200 10000000 8192,10000,65536 > $OUT/gt_fill_K100.jsonl 2> $OUT/gt_fill_K100.err
cat $OUT/gt_fill_K100.jsonl
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --configs "" --no-worstcase --no-fast --no-two-streams --cpu-seconds 0 --config1-nb 0 > $OUT/bench.json 2> $OUT/bench.err
python scripts/show_bench.py $OUT/bench.json 2>/dev/null | head -30 || tail -c 1500 $OUT/bench.json
