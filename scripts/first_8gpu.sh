#!/bin/bash
# FIRST CONTACT WITH AN 8-GPU NODE -- NEVER RUN (no round had more than one GPU).  One script, three steps, one SCALE-shaped JSON:
#   (i)   the grouped ncclSend / ncclRecv exchange between two ranks on two DEVICES (tests/test_gpu_groundtruth.py, RG_TEST_TWO_DEVICES=1;
#         on one device RCCL refuses the communicator: profiles/r05/rccl_two_process.txt)
#   (ii)  compute_groundtruth --devices 0,...,7 on 1M queries x 10M base rows (t2i shape, K = 100) with an fp64 check of a sample of rows
#   (iii) bench.py --gpus 1, 2, 4, 8 back to back -> gpurun_out/first_8gpu/SCALE.json ({"runs": [{n_gpus, value, gt_build, rc}, ...]})
# DRY_RUN=1 prints the commands (one per line, prefixed "CMD ") and runs nothing: tests/test_dist_gloo.py parses them on the CPU.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
NGPU=${NGPU:-8}; D=${D:-/tmp/first8}; O=gpurun_out/first_8gpu; B=roargraph_amd/bin
DEVS=$(seq -s, 0 $((NGPU - 1)))
export HSA_ENABLE_IPC_MODE_LEGACY=0 RG_FAULT_REPORT=$O/fault_report.txt
run() { if [ -n "$DRY_RUN" ]; then echo "CMD $*"; else echo "== $*"; "$@"; echo "== rc=$?"; fi; }
[ -n "$DRY_RUN" ] || { mkdir -p $D $O; make -s -C roargraph_amd/cli; }
# (i)
run env RG_TEST_TWO_DEVICES=1 python -m pytest tests/test_gpu_groundtruth.py -x -q -s -k test_two_processes_on_the_one_gpu_through_rg_comm_init_rank
run python -m pytest tests/test_gpu_groundtruth.py -x -q -k "single_process_multi_shard or rank_form"
# (ii)
run python scripts/first_8gpu_files.py make $D 10000000 1000000 200
run $B/compute_groundtruth --data_type float --dist_fn mips --base_file $D/base.fbin --query_file $D/query.fbin --gt_file $D/gt.bin --K 100 --devices $DEVS
run python scripts/first_8gpu_files.py check $D 64
# (iii)
for N in 1 2 4 8; do
  [ $N -le $NGPU ] || continue
  run python bench.py --gpus $N --steps 20 --warmup 5 --configs "" --full-out $O/bench_n$N.json
done
run python scripts/first_8gpu_files.py scale $O
