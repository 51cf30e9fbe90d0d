#!/bin/bash
# round-2 closing pass on the GPU box: the whole -m gpu suite, the default bench as the driver runs it, the rocprofv3 passes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_final
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
TAG=r02_final bash scripts/r02_bench.sh
L_STAR=${L_STAR:-50} bash scripts/profile_r02.sh > $OUT/profile.log 2>&1
