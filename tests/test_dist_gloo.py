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


def test_bench_plan_for_eight_gpus_at_the_10m_shape():
    """A dry run of `bench.py --gpus 8` (no 8-GPU node was available to any round so far): per-rank query
    batches, the row shards and owned query ranges of the training-query ground truth, the index broadcast, and the owner
    rows of the gt_build leg -- every partition complete and disjoint, the loads equal to within one row."""
    from roargraph_amd import dist as rgdist
    from roargraph_amd import groundtruth
    nb, nq, ntrain, gt_nq = 10_000_000, 10_000, 2_000_000, 262_144
    p = rgdist.bench_plan(8, nb, nq, ntrain, gt_nq, avg_degree=14.6)
    assert [s["first_batch_seed"] for s in p["search"]] == list(range(99, 107)) and all(s["queries_per_step"] == nq for s in p["search"])
    rows = [tuple(t["base_rows"]) for t in p["train_truth"]]
    assert rows[0][0] == 0 and rows[-1][1] == nb and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    assert {hi - lo for lo, hi in rows} == {1_250_000}
    owned = [tuple(t["owned_queries"]) for t in p["train_truth"]]
    assert owned[0][0] == 0 and owned[-1][1] == ntrain and all(a[1] == b[0] for a, b in zip(owned, owned[1:]))
    assert sum(t["scores"] for t in p["train_truth"]) == nb * ntrain                     # 2e13 scores, an eighth each
    assert all(t["all_to_all_send_bytes"] == ntrain * 100 * 8 * 7 // 8 for t in p["train_truth"])
    assert p["index_broadcast_bytes"] == (nb + 1) * 8 + int(14.6 * nb) * 4 and p["replica_bytes_per_gpu"] < 12e9
    assert sum(g["owned_rows"] for g in p["gt_build"]) == gt_nq and {g["owned_rows"] for g in p["gt_build"]} == {gt_nq // 8}
    all_rows = np.concatenate([groundtruth.owned_rows(gt_nq, 8, r, 65536) for r in range(8)])
    assert (np.sort(all_rows) == np.arange(gt_nq)).all()


@pytest.mark.parametrize("world,nq,batch", [(8, 1000, 256), (8, 65536 + 77, 0), (3, 10, 4), (8, 5, 64), (2, 200, 64), (1, 33, 16)])
def test_rccl_exchange_schedule_of_the_ground_truth_for_eight_ranks(world, nq, batch):
    """the grouped ncclSend / ncclRecv exchange of rg_groundtruth_rank has never run between two real ranks (no round
    had a multi-GPU node, RCCL refuses two ranks on one device).  The calls are issued from a table, rg_gt_exchange_plan, which needs no
    GPU: here the tables of ALL ranks of a world are played against a pure-Python model -- every send has the matching receive
    (peer, count), owners partition every batch, receive slots do not overlap and stay inside the buffer, the double-buffer
    parity alternates -- and a full exchange + merge of random K-lists through the tables returns the global top-K."""
    import ctypes as C
    from roargraph_amd._lib import check, lib
    K = 7

    def plan(rank):
        n = C.c_uint64()
        check(lib().rg_gt_exchange_plan(world, rank, C.c_uint32(nq), C.c_uint32(batch), None, C.c_uint64(0), C.byref(n)))
        out = np.zeros(n.value, np.uint32)
        check(lib().rg_gt_exchange_plan(world, rank, C.c_uint32(nq), C.c_uint32(batch), out.ctypes.data_as(C.c_void_p), C.c_uint64(n.value), C.byref(n)))
        return out.reshape(-1, 10)

    plans = [plan(r) for r in range(world)]
    Qb = min(max(nq, 1), batch or 65536)
    per = (Qb + world - 1) // world
    nbatch = (nq + Qb - 1) // Qb
    assert all(p.shape[0] == nbatch * world for p in plans)
    rng = np.random.default_rng(world * 1000 + nq)
    # the model: every rank holds K-lists (score, id) of every query over its own shard; ids are globally unique
    vals = rng.standard_normal((world, nq, K)).astype(np.float32)
    vals = -np.sort(-vals, axis=2)
    ids = (np.arange(world)[:, None, None] * 1_000_000 + rng.permutation(nq * K).reshape(nq, K)[None]).astype(np.uint32)
    out_i = np.zeros((nq, K), np.uint32); out_v = np.zeros((nq, K), np.float32); written = np.zeros(nq, np.int32)
    for b in range(nbatch):
        q0, nqb = b * Qb, min(Qb, nq - b * Qb)
        rows = [p[b * world:(b + 1) * world] for p in plans]
        # the owners' ranges: balanced, contiguous, a partition of the batch (pure-Python model of ranges_of)
        base_, extra = divmod(nqb, world)
        want_own = [(j * base_ + min(j, extra), j * base_ + min(j, extra) + base_ + (1 if j < extra else 0)) for j in range(world)]
        for r in range(world):
            assert (rows[r][:, 0] == b).all() and (rows[r][:, 1] == (b & 1)).all() and (rows[r][:, 2] == q0).all() and (rows[r][:, 3] == nqb).all()
            assert rows[r][:, 4].tolist() == list(range(world))
            for j in range(world):
                _, _, _, _, peer, s0, sn, rslot, rn, own0 = rows[r][j].tolist()
                assert (s0, s0 + sn) == want_own[j], "rank %d sends rank %d the rows it owns" % (r, j)
                assert rn == want_own[r][1] - want_own[r][0] and rslot == j * per and rslot + rn <= world * per
                assert own0 == q0 + want_own[r][0]
                # the matching call on the peer: j receives from r exactly what r sends to j
                assert rows[j][r][8] == sn and rows[j][r][4] == r
        # play the exchange: receive buffers [world * per rows], then the merge of the world lists per owned row
        for r in range(world):
            lo, hi = want_own[r]
            if hi == lo:
                continue
            rb_i = np.zeros((world * per, K), np.uint32); rb_v = np.full((world * per, K), -np.inf, np.float32)
            for j in range(world):         # what peer j sends to r lands in r's slot j
                s0, sn = rows[j][r][5], rows[j][r][6]
                slot, rn = rows[r][j][7], rows[r][j][8]
                assert sn == rn
                rb_i[slot:slot + rn] = ids[j, q0 + s0:q0 + s0 + sn]; rb_v[slot:slot + rn] = vals[j, q0 + s0:q0 + s0 + sn]
            for t in range(hi - lo):
                ci = np.concatenate([rb_i[j * per + t] for j in range(world)]); cv = np.concatenate([rb_v[j * per + t] for j in range(world)])
                order = np.lexsort((ci, -cv))[:K]
                out_i[q0 + lo + t] = ci[order]; out_v[q0 + lo + t] = cv[order]; written[q0 + lo + t] += 1
    assert (written == 1).all(), "every query row is merged by exactly one rank"
    for q in range(0, nq, max(1, nq // 50)):
        ci, cv = ids[:, q].reshape(-1), vals[:, q].reshape(-1)
        order = np.lexsort((ci, -cv))[:K]
        assert (out_i[q] == ci[order]).all() and (out_v[q] == cv[order]).all()


def test_first_8gpu_script_commands_parse():
    """scripts/first_8gpu.sh has NEVER run (no round had a multi-GPU node): what can be checked without one is that every command it
    would issue exists and parses -- bench.py's flags through bench.parse(), the ground-truth CLI's flags against its source, the pytest
    selections against the test file, the helper's sub-commands."""
    import os
    import re
    import shlex
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(["bash", os.path.join(root, "scripts", "first_8gpu.sh")], env=dict(os.environ, DRY_RUN="1"), capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    cmds = [shlex.split(l[4:]) for l in out.stdout.splitlines() if l.startswith("CMD ")]
    assert len(cmds) == 10
    sys.path.insert(0, root)
    import bench
    gpus = []
    for c in cmds:
        if "bench.py" in c:
            argv = c[c.index("bench.py") + 1:]
            if "--configs" in argv and (argv.index("--configs") + 1 == len(argv) or argv[argv.index("--configs") + 1].startswith("--")):
                argv.insert(argv.index("--configs") + 1, "")      # (the echo of the dry run drops the empty argument)
            old = sys.argv
            sys.argv = ["bench.py"] + argv
            try:
                a = bench.parse()
            finally:
                sys.argv = old
            gpus.append(a.gpus)
            assert a.steps == 20 and a.warmup == 5 and a.configs == "" and a.full_out.endswith("bench_n%d.json" % a.gpus)
        elif c[0].endswith("compute_groundtruth"):
            src = open(os.path.join(root, "roargraph_amd", "cli", "compute_groundtruth.cpp")).read()
            flags = [x[2:] for x in c if x.startswith("--")]
            assert flags and all('a.add("%s"' % f in src for f in flags), flags
            assert c[c.index("--devices") + 1] == "0,1,2,3,4,5,6,7"
        elif "pytest" in c:
            tests = open(os.path.join(root, "tests", "test_gpu_groundtruth.py")).read()
            sel = c[c.index("-k") + 1]
            for name in re.split(r"\s+or\s+", sel):
                assert re.search(r"def test_\w*%s" % re.escape(name.replace("test_", "", 1) if name.startswith("test_") else name), tests), name
        elif c[1].endswith("first_8gpu_files.py"):
            assert c[2] in ("make", "check", "scale")
    assert gpus == [1, 2, 4, 8]
    from roargraph_amd import dist as rgdist
    plan = rgdist.bench_plan(8, 10_000_000, 10_000, 2_000_000, 262_144)
    assert len(plan["search"]) == 8 and sum(b - a for a, b in (p["base_rows"] for p in plan["gt_build"])) == 10_000_000
