#!/bin/bash
# The reference's README workflow (README.md:62-117) at t2i-10M shape, literally, with the CLI twins:
#   compute_groundtruth (train queries) -> test_build_roargraph -> compute_groundtruth (test queries) -> test_search_roargraph
# over .fbin / gt / .index FILES (synthetic data of the bench's distribution written to $D first).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=${D:-/tmp/t2i}; B=roargraph_amd/bin; O=gpurun_out/e2e_cli; mkdir -p $D $O
make -s -C roargraph_amd/cli
python - <<P
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from roargraph_amd import synth, io
dev = torch.device('cuda', 0)
t = time.time()
base, train, q, desc = synth.make_device_set(dev, 1234, 10_000_000, 2_000_000, 10_000, 200, data='lowrank', rank=32, q_seed=99)
io.write_fbin('$D/base.10M.fbin', base.cpu().numpy()); io.write_fbin('$D/query.train.10M.fbin', train.cpu().numpy()); io.write_fbin('$D/query.10k.fbin', q.cpu().numpy())
print('files written in %.0f s: %s' % (time.time() - t, desc))
P
ls -la $D
T() { local what=$1; shift; local t0=$SECONDS; "$@"; echo "== $((SECONDS - t0)) s wall: $what (rc=$?)"; }
{
T "ground truth of the training queries" $B/compute_groundtruth --data_type float --dist_fn mips --base_file $D/base.10M.fbin --query_file $D/query.train.10M.fbin --gt_file $D/train.gt.bin --K 100
T "index build" $B/test_build_roargraph --data_type float --dist ip --base_data_path $D/base.10M.fbin --sampled_query_data_path $D/query.train.10M.fbin --projection_index_save_path $D/t2i_10M_roar.index --learn_base_nn_path $D/train.gt.bin --M_sq 100 --M_pjbp 35 --L_pjpq 500 -T 128 --device 0
T "ground truth of the test queries" $B/compute_groundtruth --data_type float --dist_fn mips --base_file $D/base.10M.fbin --query_file $D/query.10k.fbin --gt_file $D/gt.10k.ibin --K 100
T "search sweep" $B/test_search_roargraph --data_type float --dist ip --base_data_path $D/base.10M.fbin --projection_index_save_path $D/t2i_10M_roar.index --gt_path $D/gt.10k.ibin --query_path $D/query.10k.fbin --L_pq 10 20 30 40 50 60 80 100 150 200 300 500 1000 2000 --k 10 -T 16 --evaluation_save_path $O/eval.csv
} > $O/log.txt 2>&1
grep -v "^$" $O/log.txt | tail -40; cat $O/eval.csv; ls -la $D; rm -rf $D
