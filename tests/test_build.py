"""CPU: graph construction (rg_build_roargraph, the restated BuildRoarGraph) -- a 'next' row.

Parity status: UNPINNED.  The reference's build translation unit cannot be compiled under the rules (Boost / tsl headers
absent, stand-ins not allowed).  What stands in for a pin: TWO restatements written independently from
src/index_bipartite.cpp -- the product's builder (csrc/rg_build.cpp, with its shortcuts) and the oracle's
(oracle/rg_oracle_build.c, sweep by sweep) -- must produce the same index byte for byte, and both reproduce the md5 of the
index the survey's probe build of the reference wrote.  Plus properties and determinism.
"""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from roargraph_amd import io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rgb():
    from roargraph_amd import build
    return build


def gen(seed, nb, nt, nq, d):
    """the survey's probe generator: one rng, base then train then query"""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((nb, d)).astype(np.float32)
    train = (rng.standard_normal((nt, d)) * 0.5 + 0.3).astype(np.float32)
    query = (rng.standard_normal((nq, d)) * 0.5 + 0.3).astype(np.float32)
    return base, train, query


def np_gt(q, b, K):
    s = q.astype(np.float64) @ b.T.astype(np.float64)
    return np.argsort(-s, axis=1, kind="stable")[:, :K].astype(np.uint32)


def test_small_build_properties_and_recall(rgb, oracle):
    base, train, query = gen(5, 4000, 1500, 100, 64)
    knn = np_gt(train, base, 100)
    M = 20
    off, nbrs, ep = rgb.build_roargraph(base, knn, "ip", M_sq=100, M_pjbp=M, L_pjpq=200, num_threads=1)
    off2, nbrs2, ep2 = rgb.build_roargraph(base, knn, "ip", M_sq=100, M_pjbp=M, L_pjpq=200, num_threads=1)
    assert ep == ep2 and (off == off2).all() and (nbrs == nbrs2).all(), "one-thread build must be deterministic"
    deg = np.diff(off.astype(np.int64))
    assert deg.max() <= 2 * M and nbrs.max() < 4000
    for i in range(0, 4000, 97):
        lst = nbrs[int(off[i]):int(off[i + 1])]
        assert i not in lst and len(set(lst.tolist())) == len(lst), "self loop or repeated neighbour"
    c = base.astype(np.float64).mean(0)
    assert abs(((base[ep] - c) ** 2).sum() - ((base - c) ** 2).sum(1).min()) < 1e-3
    gt = np_gt(query, base, 100)
    ids, _, _, _ = oracle.search(base, "ip", off, nbrs, ep, query, 10, 200, nthreads=4)
    assert oracle.recall(ids, gt, 10) > 0.95
    # multi-threaded build: phase 3 in batches (a different, fixed order), the same quality
    off8, nbrs8, ep8 = rgb.build_roargraph(base, knn, "ip", M_sq=100, M_pjbp=M, L_pjpq=200, num_threads=8)
    ids8, _, _, _ = oracle.search(base, "ip", off8, nbrs8, ep8, query, 10, 200, nthreads=4)
    assert ep8 == ep and np.diff(off8.astype(np.int64)).max() <= 2 * M and oracle.recall(ids8, gt, 10) > 0.95


def structured(seed, nb, nt, d, metric, knn_k, r=6, dups=0):
    """low-rank rows (hubs: a few base points are the nearest of thousands of training queries) + optional repeated rows"""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((r, d)).astype(np.float32)
    base = rng.standard_normal((nb, r)).astype(np.float32) @ A + 0.05 * rng.standard_normal((nb, d)).astype(np.float32)
    if dups:
        base[rng.integers(0, nb, dups)] = base[rng.integers(0, nb, dups)]
    train = (rng.standard_normal((nt, r)).astype(np.float32) * 0.5 + 0.3) @ A + 0.05 * rng.standard_normal((nt, d)).astype(np.float32)
    knn = np.zeros((nt, knn_k), np.uint32)
    for i in range(0, nt, 4096):
        t = train[i:i + 4096].astype(np.float64)
        b = base.astype(np.float64)
        s = -(t @ b.T) if metric != "l2" else (t * t).sum(1)[:, None] - 2 * t @ b.T + (b * b).sum(1)[None]
        knn[i:i + 4096] = np.argsort(s, axis=1, kind="stable")[:, :knn_k]
    return base, knn


@pytest.mark.parametrize("metric,nb,nt,d,knn_k,M,L,r,dups", [
    ("ip", 3000, 20000, 16, 30, 6, 40, 6, 0),        # hubs: lists full, thousands of reverse edges into one list
    ("l2", 3000, 20000, 16, 30, 6, 40, 6, 200),      # repeated rows: ties in distance
    ("l2", 5000, 20000, 8, 30, 8, 40, 8, 0),         # full-rank: every list near its bound, ~150 windows in phase 2
    ("cosine", 2600, 6000, 24, 40, 10, 60, 8, 50),
    ("ip", 1500, 3000, 16, 20, 8, 40, 6, 0),         # <= 2,048 nodes: the schedule is all ones -> equals the one-thread build too
])
def test_many_threads_one_result_equal_to_the_oracle(rgb, oracle, metric, nb, nt, d, knn_k, M, L, r, dups):
    """rg_build_roargraph with T > 1 is deterministic (round 3): phases 1 and 2 replay every list's reverse edges in the
    one-thread order (Builder::phase1_replay, phase2_windows), phase 3 runs in the batches of rg_build_schedule over a frozen
    graph and links them in node order (link_batch).  The index must equal, byte for byte and for every thread count,
    the ORACLE's build given the same batch list (oracle/rg_oracle_build.c: plain sequential loops, searches of a batch
    before its links) -- which for a schedule of ones is the reference's one-thread sequence."""
    base, knn = structured(nb + d, nb, nt, d, metric, knn_k, r, dups)
    sched = rgb.build_schedule(nb)
    assert int(sched.sum()) == nb and (sched[: min(nb, 2048)] == 1).all()
    oracle.use_avx512(True)
    want = oracle.build_roargraph(base, knn, metric, knn_k, M, L, sched=sched)
    oracle.use_avx512(False)
    for T in (2, 5, 8):
        got = rgb.build_roargraph(base, knn, metric, knn_k, M, L, num_threads=T)
        assert got[2] == want[2], "entry point"
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all(), f"{T} threads: index differs from the oracle's"
    if nb <= 2048:
        one = rgb.build_roargraph(base, knn, metric, knn_k, M, L, num_threads=1)
        assert (one[0] == want[0]).all() and (one[1] == want[1]).all()


@pytest.mark.parametrize("metric,nb,nt,d,knn_k,M_sq,M,L", [("ip", 4000, 1500, 64, 100, 100, 20, 200), ("l2", 3000, 1000, 32, 50, 50, 16, 100),
                                                          ("cosine", 2500, 900, 40, 60, 40, 12, 80), ("ip", 1500, 700, 24, 30, 100, 35, 60)])
def test_product_build_equals_oracle_build(rgb, oracle, metric, nb, nt, d, knn_k, M_sq, M, L):
    """rg_build_roargraph at one thread against rgo_build_roargraph (the oracle's restatement of BuildRoarGraph /
    LinkProjection / the four pruning rules, :143-218, :1043-1277, :1352-1940): entry point, offsets and neighbour lists
    equal byte for byte -- inner product, L2, cosine (base normalised first, :176-182), fewer knn columns than M_sq, an M
    above what the lists reach."""
    base, train, _ = gen(nb + d, nb, nt, 10, d)
    if metric == "l2":
        s = ((train.astype(np.float64)[:, None, :] - base.astype(np.float64)[None, :, :]) ** 2).sum(-1)
        knn = np.argsort(s, axis=1, kind="stable")[:, :knn_k].astype(np.uint32)
    else:
        knn = np_gt(train, base, knn_k)
    base[7] = base[3]; base[11] = base[3]                                   # exact ties in distance
    oracle.use_avx512(True)
    want = oracle.build_roargraph(base, knn, metric, M_sq, M, L)
    oracle.use_avx512(False)
    got = rgb.build_roargraph(base, knn, metric, M_sq, M, L, num_threads=1)
    assert got[2] == want[2], "entry point"
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all(), "index differs from the oracle's"


def test_l2_build(rgb, oracle):
    base, train, query = gen(9, 3000, 1000, 80, 32)
    s = ((train.astype(np.float64)[:, None, :] - base.astype(np.float64)[None, :, :]) ** 2).sum(-1)
    knn = np.argsort(s, axis=1, kind="stable")[:, :50].astype(np.uint32)
    off, nbrs, ep = rgb.build_roargraph(base, knn, "l2", M_sq=50, M_pjbp=16, L_pjpq=100, num_threads=4)
    sq = ((query.astype(np.float64)[:, None, :] - base.astype(np.float64)[None, :, :]) ** 2).sum(-1)
    gt = np.argsort(sq, axis=1, kind="stable")[:, :100].astype(np.uint32)
    ids, _, _, _ = oracle.search(base, "l2", off, nbrs, ep, query, 10, 100, nthreads=4)
    assert np.diff(off.astype(np.int64)).max() <= 32 and oracle.recall(ids, gt, 10) > 0.9


def test_survey_probe_regression(rgb, oracle):
    """20k x 200 IP, 5k train queries, M_sq=100 M_pjbp=35 L_pjpq=500, ONE thread -- the survey's probe set.
    SURVEY.md Appendix D records for the reference's T=1 index: avg degree 41.9, max 70, recall@10 0.995 at L_pq=500,
    mean cmps 11,764.  The md5 below is the value this implementation produces; it equals the md5 of the T=1 index the
    survey's probe build wrote (a build that cannot be re-created under this round's no-stand-in rule, hence a
    regression value and an informal observation, not a parity pin)."""
    base, train, query = gen(1234, 20000, 5000, 200, 200)
    knn = np_gt(train, base, 100)
    off, nbrs, ep = rgb.build_roargraph(base, knn, "ip", M_sq=100, M_pjbp=35, L_pjpq=500, num_threads=1)
    deg = np.diff(off.astype(np.int64))
    assert abs(deg.mean() - 41.9) < 0.05 and deg.max() == 70
    import tempfile
    oracle.use_avx512(True)
    o_off, o_nbrs, o_ep = oracle.build_roargraph(base, knn, "ip", 100, 35, 500)     # the second restatement: same bytes, same md5
    oracle.use_avx512(False)
    assert o_ep == ep and (o_off == off).all() and (o_nbrs == nbrs).all()
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "t1.index")
        io.write_index(p, off, nbrs, ep)
        assert hashlib.md5(open(p, "rb").read()).hexdigest() == "f109604b40df5c87a1e52649b2de91bc"
    gt = np_gt(query, base, 100)
    ids, _, cmps, hops = oracle.search(base, "ip", off, nbrs, ep, query, 10, 500, nthreads=8)
    assert abs(oracle.recall(ids, gt, 10) - 0.995) < 0.003 and abs(cmps.mean() - 11764) < 5 and abs(hops.mean() - 501.2) < 0.1


def test_build_cli_twin(tmp_path, oracle):
    exe = os.path.join(ROOT, "roargraph_amd", "bin", "test_build_roargraph")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "roargraph_amd", "cli")])
    base, train, _ = gen(3, 1500, 400, 10, 24)
    knn = np_gt(train, base, 40)
    bf, tf, kf, out = (str(tmp_path / x) for x in ("b.fbin", "t.fbin", "knn.bin", "g.index"))
    io.write_fbin(bf, base); io.write_fbin(tf, train); io.write_gt(kf, knn)   # ids only, as LoadLearnBaseKNN accepts
    r = subprocess.run([exe, "--data_type", "float", "--dist", "ip", "--base_data_path", bf, "--sampled_query_data_path", tf,
                        "--projection_index_save_path", out, "--learn_base_nn_path", kf, "--M_sq", "40", "--M_pjbp", "12",
                        "--L_pjpq", "60", "-T", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Using inner product as distance metric" in r.stdout and "Save index to" in r.stdout
    off, nbrs, ep = oracle.index_load(out)
    from roargraph_amd import build
    o2, n2, e2 = build.build_roargraph(base, knn, "ip", 40, 12, 60, 1)
    assert ep == e2 and (off == o2).all() and (nbrs == n2).all()


def test_a_failing_worker_thread_ends_the_build_with_a_status():
    """An allocation that fails on a WORKER thread (the lists grow there) must come back as RG_ERR_OOM through the C ABI --
    not std::terminate, not team mates left in the barrier.  RG_BUILD_FAULT=n makes the n-th multi-threaded region of a
    build throw std::bad_alloc on one of its threads; every region of a small four-thread build is tried."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from roargraph_amd import build
rng = np.random.default_rng(3)
base = rng.standard_normal((1500, 32)).astype(np.float32)
train = (rng.standard_normal((600, 32)) * 0.5 + 0.3).astype(np.float32)
knn = np.argsort(-(train.astype(np.float64) @ base.T.astype(np.float64)), axis=1, kind="stable")[:, :40].astype(np.uint32)
try:
    build.build_roargraph(base, knn, "ip", M_sq=40, M_pjbp=12, L_pjpq=60, num_threads=4)
    print("BUILT")
except Exception as e:
    print("ERROR", e)
''' % ROOT
    seen_fault = seen_built = False
    for region in range(1, 40):
        env = dict(os.environ, RG_BUILD_FAULT=str(region))
        r = subprocess.run([os.sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (region, r.stdout[-300:], r.stderr[-300:])
        if "BUILT" in r.stdout:      # the build has fewer multi-threaded regions than this
            seen_built = True
            break
        assert "ERROR" in r.stdout and "out of host memory" in r.stdout, (region, r.stdout, r.stderr[-300:])
        seen_fault = True
    assert seen_fault and seen_built
