#!/bin/bash
# GPU box, round 2 first pass: the new BASELINE-shape tests, the whole -m gpu suite, the K1 phase profile on a genuine
# index (instrumented build), the same searches with the product build, and rocprofv3 counters on the genuine-index launch.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_box1
mkdir -p $OUT
cd $R
NB=${NB:-2000000}
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -m gpu > $OUT/tests_shapes.log 2>&1; echo "shapes rc=$?" >> $OUT/tests_shapes.log
timeout 600 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_baseline_shapes.py > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
RG_HIP_LIB=$R/roargraph_amd/librg_hip_prof.so timeout 900 python scripts/exp/k1_phases.py --nb $NB --save /tmp/ix --out $OUT/k1_phases_prof.json > $OUT/k1_phases_prof.log 2>&1
timeout 600 python scripts/exp/k1_phases.py --nb $NB --load /tmp/ix --out $OUT/k1_phases_product.json > $OUT/k1_phases_product.log 2>&1
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args..., -- command
  local name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 "$@" > $OUT/$name.log 2>&1
  local db=$(ls /tmp/rp_$name/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocprof_summary.py --json $OUT/$name.pmc.json $db > $OUT/$name.txt 2>&1; fi
  rm -rf /tmp/rp_$name
  grep -v "simple_timer\|SQLite3" $OUT/$name.log | tail -8 > $OUT/$name.log.tail; rm -f $OUT/$name.log
}
for L in 500 2000; do
  for M in 1 0; do
    B="python $R/scripts/exp/k1_phases.py --nb $NB --load /tmp/ix --Ls $L --modes $M"
    run pmc_sq1_L${L}_m$M --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d /tmp/rp_pmc_sq1_L${L}_m$M -o s -- $B
    run pmc_fetch_L${L}_m$M --pmc FETCH_SIZE -d /tmp/rp_pmc_fetch_L${L}_m$M -o s -- $B
  done
done
ls -la $OUT
