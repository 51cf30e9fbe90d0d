#!/bin/bash
# round 4, box 5: the exact visited set in LDS (K1 VIS = 3): parity, then A/B on the 10M bench index
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box5
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact_set_in_lds or repeats" > $OUT/pytest_new.log 2>&1
tail -15 $OUT/pytest_new.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_concurrency.py tests/test_gpu_golden.py tests/test_gpu_cli.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
RG_TRACE_ADAPTIVE=1 timeout 1500 python scripts/exp/k1_ab.py --L 10,20,30,50,60,80,100,150 --index-cache /tmp/ix.npz --pipelined \
  --configs "default:visited=2;nolset:visited=2,lset=0;lset_forced:visited=2,lset=100000;filter:visited=1;filter_w12:visited=1,waves_per_cu=12;filter_w10:visited=1,waves_per_cu=10" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep -c "exact LDS set" $OUT/k1_ab.err; grep "exact LDS set" $OUT/k1_ab.err | sort | uniq -c | sort -rn | head -20
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['config'], r['L'], r['pct_of_8TBs'], r['same_ids_hops'], r['same_cmps'])"
