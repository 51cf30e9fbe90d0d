"""-m gpu: K2/K3 (fp32 MFMA brute-force top-K) against the fp64 oracle.

The MFMA sums in a different order than any CPU GEMM, so ids are compared with a tie band:
every returned id must be a legitimate member of the top-K up to 1e-5 relative in the fp64 score, the returned order
must follow the fp64 scores up to that band, and the stored distances must be within 1e-4 relative of fp64.
"""
import numpy as np
import pytest

from roargraph_amd import synth

pytestmark = pytest.mark.gpu


def fp64_scores(base, q, ids, metric):
    b = base.astype(np.float64)[ids.astype(np.int64)]          # [nq, K, d]
    qq = q.astype(np.float64)[:, None, :]
    if metric == "l2":
        return ((qq - b) ** 2).sum(-1)
    return (qq * b).sum(-1)


def check_gt(base, q, metric, K, ids, dists, ref_ids, ref_s64, tol=1e-5, min_same=0.999):
    nq = q.shape[0]
    s = fp64_scores(base, q, ids, metric)
    scale = np.abs(ref_s64).max(axis=1, keepdims=True) + 1e-30
    for i in range(nq):
        assert len(set(ids[i].tolist())) == K, "duplicate ids in row %d" % i
    if metric == "l2":
        assert (s <= ref_s64[:, -1:] + tol * scale).all(), "an id outside the true top-K was returned"
        assert (np.diff(s, axis=1) >= -tol * scale).all(), "rows not sorted best-first"
    else:
        assert (s >= ref_s64[:, -1:] - tol * scale).all(), "an id outside the true top-K was returned"
        assert (np.diff(s, axis=1) <= tol * scale).all(), "rows not sorted best-first"
    # rank-by-rank: same id, or a tie-band neighbour
    same = ids == ref_ids
    assert (np.abs(s - ref_s64)[~same] <= tol * np.broadcast_to(scale, s.shape)[~same]).all()
    assert same.mean() > min_same
    assert (np.abs(dists.astype(np.float64) - s) <= 1e-4 * np.abs(s) + 1e-4 * scale * 1e-2).all(), "stored distances off"


@pytest.mark.parametrize("metric,d,nb,nq,K", [("ip", 200, 20000, 300, 100), ("l2", 512, 6000, 130, 100), ("ip", 512, 9000, 200, 100),
                                              ("ip", 200, 1000, 5, 10), ("l2", 24, 3000, 257, 1),
                                              ("ip", 104, 5000, 64, 100), ("ip", 200, 130, 128, 128),
                                              ("l2", 200, 4000, 100, 300),
                                              # the other register-stationary instantiations (common embedding widths)
                                              ("l2", 96, 9000, 200, 100), ("ip", 128, 9000, 200, 100),
                                              ("l2", 256, 5000, 150, 100), ("ip", 384, 5000, 150, 50)])
def test_groundtruth_vs_fp64(oracle, metric, d, nb, nq, K):
    from roargraph_amd import groundtruth
    base, q = synth.make_synth(77, nb, nq, d)
    ids, dists = groundtruth.compute_groundtruth(base, q, metric, K)
    ref_ids, _, ref_s = oracle.groundtruth_f64(base, q, metric, K, nthreads=16)
    check_gt(base, q, metric, K, ids, dists, ref_ids, ref_s)


def test_groundtruth_exact_ties_by_id(oracle):
    """Duplicated base rows give exactly equal scores: order must fall back to id ascending."""
    from roargraph_amd import groundtruth
    base, q = synth.make_synth(5, 2000, 50, 200)
    base[1000:2000] = base[0:1000]
    ids, dists = groundtruth.compute_groundtruth(base, q, "ip", 20)
    ref_ids, _, ref_s = oracle.groundtruth_f64(base, q, "ip", 20, nthreads=8)
    assert (ids[:, 0::2] + 1000 == ids[:, 1::2]).all(), "equal scores must be ordered by id"
    check_gt(base, q, "ip", 20, ids, dists, ref_ids, ref_s)


def test_shard_merge_equals_single(oracle):
    """Base split in 3 row shards (K2 with id_base) + K3 merge == one-shot result, bit for bit."""
    import torch
    from roargraph_amd import groundtruth
    base, q = synth.make_synth(9, 9000, 200, 200)
    K = 100
    dev = torch.device("cuda", 0)
    bt, qt = torch.from_numpy(base).to(dev), torch.from_numpy(q).to(dev)
    one_i = torch.zeros((200, K), dtype=torch.int32, device=dev)
    one_v = torch.zeros((200, K), dtype=torch.float32, device=dev)
    groundtruth.gt_shard_dev(bt, qt, "ip", K, 0, one_i, one_v)
    parts_i = torch.zeros((3, 200, K), dtype=torch.int32, device=dev)
    parts_v = torch.zeros((3, 200, K), dtype=torch.float32, device=dev)
    for r, (lo, hi) in enumerate(groundtruth.shard_rows(9000, 3)):
        groundtruth.gt_shard_dev(bt[lo:hi], qt, "ip", K, lo, parts_i[r], parts_v[r])
    mi = torch.zeros_like(one_i)
    mv = torch.zeros_like(one_v)
    groundtruth.gt_merge_dev(parts_i, parts_v, 3, 200, K, "ip", mi, mv)
    torch.cuda.synchronize()
    assert torch.equal(mi, one_i)
    assert torch.equal(mv.view(torch.int32), one_v.view(torch.int32))


def test_gt_file_roundtrip(tmp_path, oracle):
    """CLI-body form: .fbin in, gt file out, readable by the reference's loader rules (ids block + dists block)."""
    from roargraph_amd import groundtruth, index, io
    base, q = synth.make_synth(3, 3000, 40, 200)
    io.write_fbin(str(tmp_path / "b.fbin"), base)
    io.write_fbin(str(tmp_path / "q.fbin"), q)
    groundtruth.compute_groundtruth_files(str(tmp_path / "b.fbin"), str(tmp_path / "q.fbin"), str(tmp_path / "gt.bin"), "ip", 100)
    assert oracle.gt_meta(str(tmp_path / "gt.bin")) == (40, 100)
    ids, ds = index.gt_load(str(tmp_path / "gt.bin"))
    ref_ids, _, ref_s = oracle.groundtruth_f64(base, q, "ip", 100, nthreads=8)
    check_gt(base, q, "ip", 100, ids, ds, ref_ids, ref_s)
    assert (index.knn_ids_load(str(tmp_path / "gt.bin")) == ids).all()


def test_single_process_multi_shard_path(oracle):
    """rg_groundtruth_mem with several entries in `devices` shards the base by rows, one shard per entry, and merges with
    K3.  On a one-GPU box the same device is listed three times: three shards, three streams, one merge."""
    from roargraph_amd import groundtruth
    base, q = synth.make_synth(41, 7001, 150, 200)
    one_i, one_d = groundtruth.compute_groundtruth(base, q, "ip", 50)
    three_i, three_d = groundtruth.compute_groundtruth(base, q, "ip", 50, devices=[0, 0, 0])
    assert (one_i == three_i).all() and (one_d.view(np.uint32) == three_d.view(np.uint32)).all()
    l2_i, l2_d = groundtruth.compute_groundtruth(base, q, "l2", 10, devices=[0, 0])
    ref_i, _, ref_s = oracle.groundtruth_f64(base, q, "l2", 10, nthreads=8)
    check_gt(base, q, "l2", 10, l2_i, l2_d, ref_i, ref_s)


@pytest.mark.parametrize("metric", ["ip", "l2", "cosine"])
def test_streamed_batches_equal_one_shot(oracle, metric, monkeypatch):
    """rg_groundtruth_mem streams the queries in batches (RG_GT_BATCH forces small ones: 7 batches here, the last one
    short) through the double-buffered K2 / exchange / K3 / download pipeline; the result equals the one-batch run bit for
    bit, on one rank and on three ranks sharing the GPU (in-process transport)."""
    from roargraph_amd import groundtruth
    base, q = synth.make_synth(61, 6000, 333, 200)
    monkeypatch.delenv("RG_GT_BATCH", raising=False)
    one_i, one_d = groundtruth.compute_groundtruth(base, q, metric, 64)
    monkeypatch.setenv("RG_GT_BATCH", "50")
    for devs in ([0], [0, 0, 0]):
        got_i, got_d = groundtruth.compute_groundtruth(base, q, metric, 64, devices=devs)
        assert (got_i == one_i).all() and (got_d.view(np.uint32) == one_d.view(np.uint32)).all(), devs
    if metric == "cosine":   # fp64 truth of the normalised rows (compute_groundtruth --dist_fn cosine: IP on unit vectors)
        nb_, nq_ = base.copy(), q.copy()
        oracle.normalize_rows(nb_)
        oracle.normalize_rows(nq_)
        ref_i, _, ref_s = oracle.groundtruth_f64(nb_, nq_, "ip", 64, nthreads=8)
        check_gt(nb_, nq_, "ip", 64, one_i, one_d, ref_i, ref_s)
    else:
        ref_i, _, ref_s = oracle.groundtruth_f64(base, q, metric, 64, nthreads=8)
        check_gt(base, q, metric, 64, one_i, one_d, ref_i, ref_s)


def test_rank_form_owned_rows_and_rccl_loopback(oracle, monkeypatch):
    """rg_groundtruth_rank through its own handles: three in-process ranks (threads) fill disjoint rows of one output
    array; and a single rank forced through RCCL (ncclCommInitAll on the one GPU, send/recv to itself) -- the dlopen'ed
    library, the grouped send/recv and the stream/event hand-over are the ones a multi-GPU run uses."""
    import threading
    import torch
    from roargraph_amd import groundtruth
    base, q = synth.make_synth(62, 5000, 210, 200)
    K, batch = 32, 64
    want_i, want_d = groundtruth.compute_groundtruth(base, q, "ip", K)
    dev = torch.device("cuda", 0)
    comms = groundtruth.Comm.local([0, 0, 0])
    assert not any(c.uses_rccl() for c in comms)            # ranks sharing a device: peer copies
    out_i = np.zeros((210, K), np.uint32); out_d = np.zeros((210, K), np.float32)
    shards = groundtruth.shard_rows(5000, 3)
    errs = []

    def work(r):
        try:
            lo, hi = shards[r]
            bt = torch.from_numpy(base[lo:hi]).to(dev)
            groundtruth.groundtruth_rank(comms[r], bt, lo, q, "ip", K, out_i, out_d, batch=batch)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(r,)) for r in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    [c.destroy() for c in comms]
    assert not errs, errs
    assert (out_i == want_i).all() and (out_d.view(np.uint32) == want_d.view(np.uint32)).all()
    rows = np.concatenate([groundtruth.owned_rows(210, 3, r, batch) for r in range(3)])
    assert sorted(rows.tolist()) == list(range(210))          # the three ranks' rows partition the job
    # one rank, RCCL transport
    monkeypatch.setenv("RG_GT_FORCE_RCCL", "1")
    (c,) = groundtruth.Comm.local([0])
    if not c.uses_rccl():
        c.destroy()
        pytest.skip("librccl could not be loaded in this process")
    out_i[:] = 0; out_d[:] = 0
    groundtruth.groundtruth_rank(c, torch.from_numpy(base).to(dev), 0, q, "ip", K, out_i, out_d, batch=batch)
    c.destroy()
    assert (out_i == want_i).all() and (out_d.view(np.uint32) == want_d.view(np.uint32)).all()


def test_rank_form_a_failing_rank_releases_its_peers():
    """Three in-process ranks (threads) through rg_comm_init_local + rg_groundtruth_rank, the public path.  Rank 2's shard has
    fewer rows than K, a precondition that fails on that rank only: the other two must come back with an error instead of
    waiting for it in the exchange barrier for ever, and tearing the group down afterwards must be safe."""
    import threading
    import torch
    from roargraph_amd import groundtruth
    base, q = synth.make_synth(63, 3000, 150, 200)
    K, batch = 40, 64
    dev = torch.device("cuda", 0)
    comms = groundtruth.Comm.local([0, 0, 0])
    out_i = np.zeros((150, K), np.uint32); out_d = np.zeros((150, K), np.float32)
    shards = [(0, 1500), (1500, 2980), (2980, 3000)]          # 20 rows < K on rank 2
    errs = [None] * 3

    def work(r):
        try:
            lo, hi = shards[r]
            bt = torch.from_numpy(base[lo:hi]).to(dev)
            groundtruth.groundtruth_rank(comms[r], bt, lo, q, "ip", K, out_i, out_d, batch=batch)
        except Exception as e:  # noqa: BLE001
            errs[r] = str(e)

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(3)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th), "a rank is still waiting for the one that failed"
    assert errs[2] is not None and "K must be in" in errs[2]
    assert errs[0] is not None and errs[1] is not None and all("peer rank failed" in e for e in errs[:2]), errs
    [c.destroy() for c in comms]


@pytest.mark.parametrize("metric,d,K", [("ip", 200, 100), ("l2", 512, 100), ("ip", 24, 10)])
def test_balanced_work_split_equals_the_equal_items_form(oracle, metric, d, K, monkeypatch):
    """Round 4: query counts whose blocks would need a partial second round of the resident workgroups (here 80,000 queries =
    625 blocks for 512 / 256 of them) run as one equal stretch of the (query block x base tile) rectangle per workgroup -- a
    block's top-K comes from up to 1024 / K pieces merged by K3.  Same lists as the older form (equal items handed out by a
    counter), bit for bit in ids, and both against fp64 on a sample."""
    import torch
    from roargraph_amd import groundtruth
    dev = torch.device("cuda", 0)
    nb = 100_000
    base = synth.make_synth(31, nb, 16, d)[0]
    tb = torch.from_numpy(base).to(dev)
    # 80,000 queries: the balanced split; 300 queries: three blocks cut into ten row segments each (the equal-items form either way)
    for nq in (80_000, 300):
        q = synth.make_synth(32, 16, nq, d)[1]
        tq = torch.from_numpy(q).to(dev)
        res = {}
        for name, env in (("default", {}), ("equal_items", {"RG_GT_NOBALANCE": "1"})):
            for k_ in ("RG_GT_NOBALANCE",):
                monkeypatch.delenv(k_, raising=False)
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            ids = torch.zeros((nq, K), dtype=torch.int32, device=dev); vals = torch.zeros((nq, K), device=dev)
            groundtruth.gt_shard_dev(tb, tq, metric, K, 0, ids, vals); torch.cuda.synchronize()
            res[name] = (ids.cpu().numpy().view(np.uint32), vals.cpu().numpy())
        for name in ("equal_items",):
            assert (res["default"][0] == res[name][0]).all(), (name, nq)
            assert (res["default"][1].view(np.uint32) == res[name][1].view(np.uint32)).all(), (name, nq)
        sel = np.arange(0, nq, 331 if nq > 1000 else 7)
        ref_ids, _, ref_s = oracle.groundtruth_f64(base, q[sel], metric, K, nthreads=16)
        check_gt(base, q[sel], metric, K, res["default"][0][sel], res["default"][1][sel], ref_ids, ref_s, min_same=0.997)



def test_two_processes_on_the_one_gpu_through_rg_comm_init_rank(tmp_path):
    """8-GPU readiness that one GPU can show: two PROCESSES join one communicator through rg_comm_unique_id /
    rg_comm_init_rank -- the path `compute_groundtruth` under a launcher and bench.py --gpus N take -- both on device 0.
    RCCL may refuse a communicator with two ranks on one device; whatever it does, both ranks must come back (no hang), with
    the same verdict, and where the communicator exists the streamed ground truth over it must equal the one-process result.
    The outcome is printed (pytest -s / the captured log) so that the record says which of the two this runtime did."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes as C, os, sys, time, numpy as np
sys.path.insert(0, %r)
rank = int(sys.argv[1]); idfile = sys.argv[2]; out = sys.argv[3]
import torch
from roargraph_amd import groundtruth, synth
from roargraph_amd._lib import lib
L = lib()
if rank == 0:
    buf = C.create_string_buffer(128)
    rc = L.rg_comm_unique_id(buf)
    open(idfile + ".tmp", "wb").write(buf.raw if rc == 0 else b"")
    os.rename(idfile + ".tmp", idfile)
t0 = time.time()
while not os.path.exists(idfile):
    if time.time() - t0 > 60: print("RESULT rank %%d: no id file" %% rank); sys.exit(0)
    time.sleep(0.05)
ident = open(idfile, "rb").read()
if not ident:
    print("RESULT rank %%d: rccl unavailable (rg_comm_unique_id failed)" %% rank); sys.exit(0)
h = C.c_void_p()
# RG_TEST_TWO_DEVICES=1 (scripts/first_8gpu.sh, a node with >= 2 GPUs): rank r on device r -- the exchange between two DEVICES
device = rank if (os.environ.get("RG_TEST_TWO_DEVICES") and torch.cuda.device_count() >= 2) else 0
torch.cuda.set_device(device)
rc = L.rg_comm_init_rank(ident, rank, 2, device, C.byref(h))
if rc != 0:
    print("RESULT rank %%d: init refused: %%s" %% (rank, L.rg_last_error().decode())); sys.exit(0)
comm = groundtruth.Comm(h, rank, 2, device)
base, q = synth.make_synth(64, 4000, 200, 200)
lo, hi = groundtruth.shard_rows(4000, 2)[rank]
oi = np.zeros((200, 24), np.uint32); od = np.zeros((200, 24), np.float32)
groundtruth.groundtruth_rank(comm, torch.from_numpy(base[lo:hi]).cuda(), lo, q, "ip", 24, oi, od, batch=64)
np.savez(out, ids=oi, dists=od)
comm.destroy()
print("RESULT rank %%d: ok rccl=%%d" %% (rank, 1))
''' % root
    idfile = str(tmp_path / "nccl_id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), idfile, str(tmp_path / ("out%d.npz" % r))], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for pp in procs:
                pp.kill()
            pytest.fail("a rank hung in the two-process communicator set-up")
        outs.append(o)
    res = [[l for l in o.splitlines() if l.startswith("RESULT")] for o in outs]
    print("\n".join(sum(res, [])))
    # the record of what this runtime did: kept under gpurun_out/ (merged back by gpurun), copied to profiles/r05/
    try:
        rec_dir = os.path.join(root, "gpurun_out")
        os.makedirs(rec_dir, exist_ok=True)
        import torch
        with open(os.path.join(rec_dir, "rccl_two_process.txt"), "w") as fh:
            fh.write("two processes, rg_comm_unique_id + rg_comm_init_rank(rank, world = 2, device 0) on a box with %d GPU(s); torch %s, HIP %s\n"
                     % (torch.cuda.device_count(), torch.__version__, torch.version.hip))
            fh.write("\n".join(sum(res, [])) + "\n")
            for r_, o in enumerate(outs):
                extra = [l for l in o.splitlines() if not l.startswith("RESULT") and ("NCCL" in l or "RCCL" in l or "rccl" in l or "nccl" in l)][-6:]
                fh.write("".join("rank %d output: %s\n" % (r_, l) for l in extra))
    except OSError:
        pass
    assert all(len(r) == 1 for r in res), outs
    ok = ["ok rccl" in r[0] for r in res]
    assert ok[0] == ok[1], res      # both joined, or both were refused
    if all(ok):
        from roargraph_amd import groundtruth
        base, q = synth.make_synth(64, 4000, 200, 200)
        want_i, want_d = groundtruth.compute_groundtruth(base, q, "ip", 24)
        got_i = np.zeros_like(want_i); got_d = np.zeros_like(want_d)
        for r in range(2):
            z = np.load(str(tmp_path / ("out%d.npz" % r)))
            rows = groundtruth.owned_rows(200, 2, r, 64)
            got_i[rows] = z["ids"][rows]; got_d[rows] = z["dists"][rows]
        assert (got_i == want_i).all() and (got_d.view(np.uint32) == want_d.view(np.uint32)).all()
    else:
        assert all(("refused" in r[0]) or ("unavailable" in r[0]) for r in res), res


@pytest.mark.parametrize("metric,d,K", [("ip", 200, 100), ("l2", 200, 100), ("ip", 512, 10), ("ip", 200, 128)])
@pytest.mark.parametrize("order", ["best_last", "best_first", "best_in_one_piece"])
def test_quota_thresholds_when_one_piece_holds_every_neighbour(oracle, metric, d, K, order, monkeypatch):
    """Round 5: a query block whose rows are searched in pieces filters every piece with min_j u_j, u_j = piece j's ceil(K rows_j / nb)-th
    best -- a lower bound of the shard's K-th best.  The adversarial layouts: the base sorted so that ONE piece holds all true
    neighbours (the other pieces end with fewer than K entries and pad their lists; K3 must still return the exact top-K), sorted
    the other way round (the first tiles set a threshold nothing later beats), and every tile better than all before it (all 128
    rows of a tile pass for every query: the candidate buffers fill at the highest rate).  Few queries, so that the segment form
    cuts every block into many pieces; the lists equal the fp64 brute force and do not depend on RG_GT_NOSHARE."""
    from roargraph_amd import groundtruth
    rng = np.random.default_rng(K + d)
    nb, nq = 48_000, 200
    base = rng.standard_normal((nb, d)).astype(np.float32)
    q = (rng.standard_normal((nq, d)) * 0.2 + 1.0).astype(np.float32)      # queries near (1, ..., 1): the score grows with the row sum
    key = base.sum(axis=1) if metric == "ip" else -np.linalg.norm(base - 1.0, axis=1)
    idx = np.argsort(key)
    if order == "best_first":
        idx = idx[::-1]
    elif order == "best_in_one_piece":
        top = idx[-6000:]; rest = rng.permutation(idx[:-6000])
        idx = np.concatenate([rest[:20000], top, rest[20000:]])
    base = np.ascontiguousarray(base[idx])
    ids, dists = groundtruth.compute_groundtruth(base, q, metric, K)
    ref_ids, _, ref_s = oracle.groundtruth_f64(base, q, metric, K, nthreads=16)
    check_gt(base, q, metric, K, ids, dists, ref_ids, ref_s)
    monkeypatch.setenv("RG_GT_NOSHARE", "1")
    ids2, dists2 = groundtruth.compute_groundtruth(base, q, metric, K)
    assert (ids2 == ids).all() and (dists2.view(np.uint32) == dists.view(np.uint32)).all()


def test_ties_across_the_pieces_of_a_query_block_keep_the_smaller_id(oracle, monkeypatch):
    """the bound the pieces of a query block share (min_j u_j, quota thresholds) reaches a piece's filter as a STRICT threshold.
    Duplicated base rows in different pieces score exactly alike; if the bound equals that score, the piece that hears of it first would
    drop its twin -- possibly the one with the SMALLER id -- and the result would depend on timing.  The shared bound is therefore the
    next float below min_j u_j.  48,000 rows in pieces of a few thousand, every row duplicated 24,000 rows further on, few queries:
    equal scores must come out id-ascending, and the lists must not depend on RG_GT_NOSHARE (every piece on its own) -- five times."""
    from roargraph_amd import groundtruth
    rng = np.random.default_rng(77)
    nb, nq, K, d = 48_000, 96, 100, 200
    half = rng.standard_normal((nb // 2, d)).astype(np.float32)
    base = np.concatenate([half, half])
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ref_ids, _, ref_s = oracle.groundtruth_f64(base, q, "ip", K, nthreads=16)
    runs = []
    for _ in range(5):
        ids, dists = groundtruth.compute_groundtruth(base, q, "ip", K)
        assert (ids[:, 0::2] + nb // 2 == ids[:, 1::2]).all(), "equal scores must be ordered by id"
        assert (dists[:, 0::2].view(np.uint32) == dists[:, 1::2].view(np.uint32)).all()
        runs.append(ids)
    assert all((r == runs[0]).all() for r in runs)
    check_gt(base, q, "ip", K, runs[0], dists, ref_ids, ref_s)
    monkeypatch.setenv("RG_GT_NOSHARE", "1")
    ids2, dists2 = groundtruth.compute_groundtruth(base, q, "ip", K)
    assert (ids2 == runs[0]).all() and (dists2.view(np.uint32) == dists.view(np.uint32)).all()
