#!/bin/bash
# kernel-trace of the 10M build inside bench.py: how the GPU part of phase 3 splits between the searches and the pruning
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
o=gpurun_out/build_prof; mkdir -p $o; rm -rf /tmp/rp_build
RG_BUILD_TIMING=1 rocprofv3 --kernel-trace --stats -d /tmp/rp_build -o s -- python bench.py --steps 2 --warmup 1 --sweep= --L 50 --no-worstcase --no-fast --no-two-streams --gt-nq 0 --config1-nb 0 --cpu-seconds 0 > $o/log.txt 2>&1
db=$(ls /tmp/rp_build/*.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $db > $o/build_trace.txt 2>&1
grep rg_build $o/log.txt; head -14 $o/build_trace.txt | cut -c1-170
