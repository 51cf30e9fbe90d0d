#!/bin/bash
# Per-hop instruction/cycle budget of K1 on a navigable graph (recall_curve workload, L_pq from $1, default 500).
R=${GRAFT_REPO_ROOT:-/root/repo}; L=${1:-500}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_hop
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  -d /tmp/rp_hop -o s -- python $R/scripts/recall_curve.py --L $L > /tmp/rp_hop.log 2>&1
grep '^{"L_pq"' /tmp/rp_hop.log
python $R/scripts/rocprof_summary.py /tmp/rp_hop/*.db | grep -E "rg_search_kernel|rg_distinct" | cut -c1-60,70-140
