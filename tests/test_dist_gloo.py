"""CPU, world_size 2 over gloo: the N>1 orchestration (partitioning, id offsets, all-to-all layout, merge, gather).

The per-rank compute (K1/K2/K3) needs a GPU, so these tests inject numpy stand-ins built on the oracle for it --
test doubles, passed explicitly; the product has no CPU fallback.  What is under test is everything around the kernels.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _np_shard(base_shard, queries, metric, K, id_base, ids, vals):
    from oracle import pyoracle as po
    i, d, _ = po.groundtruth_f64(base_shard.numpy(), queries.numpy(), metric, K, nthreads=2)
    ids.copy_(torch.from_numpy((i + id_base).astype(np.int32)))
    vals.copy_(torch.from_numpy(d))


def _np_merge(recv_i, recv_v, nlists, nq, K, metric, out_i, out_v):
    i = recv_i.numpy().astype(np.int64)
    v = recv_v.numpy().astype(np.float64)
    for q in range(nq):
        ci, cv = i[:, q].reshape(-1), v[:, q].reshape(-1)
        order = np.lexsort((ci, cv if metric == "l2" else -cv))[:K]
        out_i[q] = torch.from_numpy(ci[order].astype(np.int32))
        out_v[q] = torch.from_numpy(cv[order].astype(np.float32))


def _worker(rank, world, port, metric, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    from roargraph_amd import dist as rgdist
    from roargraph_amd import groundtruth, synth
    nb, nq, d, K = 1501, 37, 24, 10
    base, q = synth.make_synth(3, nb, nq, d)
    lo, hi = groundtruth.shard_rows(nb, world)[rank]
    ids, vals = groundtruth.groundtruth_distributed(torch.from_numpy(base[lo:hi].copy()), lo, torch.from_numpy(q), metric,
                                                    K, shard_fn=_np_shard, merge_fn=_np_merge)
    ref_i, ref_d, _ = po.groundtruth_f64(base, q, metric, K, nthreads=2)
    qlo, qhi = groundtruth.query_ranges(nq, world)[rank]
    ok_gt = bool((ids.numpy().astype(np.uint32) == ref_i[qlo:qhi]).all()) and bool(np.allclose(vals.numpy(), ref_d[qlo:qhi]))

    # query-sharded search: every rank searches its slice with a stand-in search_fn, results gathered everywhere
    lists, ep = synth.knn_graph(base, metric, M=6)
    from roargraph_amd import io
    off, nbrs = io.lists_to_csr(lists)

    def fake_search(qs, k, L):
        return po.search(base, metric, off, nbrs, ep, qs, k, L, nthreads=1)

    full = rgdist.search_sharded(fake_search, q, 5, 20, gather=True)
    want = po.search(base, metric, off, nbrs, ep, q, 5, 20, nthreads=1)
    ok_s = all(bool((a == b).all()) for a, b in zip(full, want))
    ret[rank] = (ok_gt, ok_s)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_world2_gloo(metric):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), metric, ret), nprocs=world, join=True)
    assert dict(ret) == {0: (True, True), 1: (True, True)}


def test_partition_helpers():
    from roargraph_amd import dist as rgdist
    from roargraph_amd import groundtruth
    for n in (0, 1, 7, 8, 9, 10000):
        for w in (1, 2, 3, 8):
            r = groundtruth.shard_rows(n, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert [rgdist.query_slice(n, i, w) for i in range(w)] == groundtruth.query_ranges(n, w)


def test_row_shards_are_balanced_and_owned_rows_partition_the_job():
    """shard_rows: floor(n/w) rows each, the first n % w one more -- no shard shorter than K or empty when n is not a
    multiple of w (810 rows over 8 ranks with K = 100 used to leave 96 in the last one).  owned_rows mirrors the batching
    of rg_groundtruth_rank: per batch, rank r owns the r-th balanced slice; together the ranks own every row once."""
    import numpy as np
    from roargraph_amd import groundtruth
    for n, w in ((810, 8), (9, 4), (10_000_000, 8), (5, 5), (1000, 3)):
        sizes = [hi - lo for lo, hi in groundtruth.shard_rows(n, w)]
        assert sum(sizes) == n and max(sizes) - min(sizes) <= 1 and min(sizes) == n // w
    for nq, w, batch in ((210, 3, 64), (65536 * 2 + 5, 8, 0), (7, 2, 100), (64, 4, 64)):
        rows = np.concatenate([groundtruth.owned_rows(nq, w, r, batch) for r in range(w)])
        assert rows.shape[0] == nq and (np.sort(rows) == np.arange(nq)).all()
