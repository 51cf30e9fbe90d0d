"""CPU: the C oracle against the committed golden vectors (tests/golden/, produced by scripts/make_golden.py from the
reference's own headers).  This is what pins the oracle on machines that have no reference tree."""
import glob
import os

import numpy as np
import pytest

from helpers import bits

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "dist_*.npz"))))
def test_distance_goldens(oracle, path):
    z = np.load(path)
    metric = os.path.basename(path).split("_")[1]
    for avx in (False, True):
        oracle.use_avx512(avx)
        got = oracle.compare_pairs(metric, z["a"], z["b"])
        oracle.use_avx512(False)
        assert (bits(got) == z["expect_bits"]).all(), (path, avx)


def test_queue_goldens(oracle):
    z = np.load(os.path.join(GOLD, "queue_traces.npz"))
    for t in range(int(z["ntraces"])):
        r = oracle.queue_trace(int(z["t%d_cap" % t]), z["t%d_ops" % t], z["t%d_ids" % t], z["t%d_dists" % t])
        assert r["size"] == int(z["t%d_size" % t]) and r["cur"] == int(z["t%d_cur" % t])
        assert (r["ids"] == z["t%d_out_ids" % t]).all()
        assert (bits(r["dists"]) == z["t%d_out_dists" % t]).all()
        assert (r["flags"] == z["t%d_out_flags" % t]).all()
        npop = z["t%d_pops" % t].shape[0]
        assert (r["pops"][:npop] == z["t%d_pops" % t]).all()


@pytest.mark.parametrize("name", ["ip200", "l2_512", "ip24"])
def test_search_goldens(oracle, name):
    z = np.load(os.path.join(GOLD, "search_%s.npz" % name))
    metric = str(z["metric"])
    for tag in z["configs"]:
        L, k = (int(x[1:]) for x in str(tag).split("_"))
        ids, ds, cmps, hops = oracle.search(z["base"], metric, z["offsets"], z["nbrs"], int(z["ep"]), z["queries"], k, L,
                                            nthreads=2)
        assert (ids == z[tag + "_ids"]).all(), tag
        assert (bits(ds) == z[tag + "_dist_bits"]).all(), tag
        assert (cmps == z[tag + "_cmps"]).all() and (hops == z[tag + "_hops"]).all(), tag


def cosine_close(ids, ds, cmps, hops, z, tag):
    """Cosine against the reference build's golden (scripts/make_golden.py g3c).  The reference's normalize<float>
    compiles to vrsqrtss + one Newton step (an estimate whose bits differ between CPU vendors), so its cosine results are
    not a bit-level contract: ids / cmps / hops must be equal (the fixture's rank gaps are far above the rounding), the
    distances within the north star's 1e-4 relative -- in fact within 4 ulp."""
    assert (ids == z[tag + "_ids"]).all() and (cmps == z[tag + "_cmps"]).all() and (hops == z[tag + "_hops"]).all(), tag
    want = z[tag + "_dist_bits"].view(np.float32)
    assert np.abs((ds - want) / want).max() <= 2e-6, tag


def test_search_golden_cosine(oracle):
    """a3: COSINE = normalize<float> over base rows and queries (util.h:214-225; index_bipartite.cpp:2679-2684;
    test_search_roargraph.cpp:167-172), then the IP kernel (index.cpp:8-26)."""
    z = np.load(os.path.join(GOLD, "search_cos200.npz"))
    b, q = z["base"].copy(), z["queries"].copy()
    oracle.normalize_rows(b)
    oracle.normalize_rows(q)
    for tag in z["configs"]:
        L, k = (int(x[1:]) for x in str(tag).split("_"))
        ids, ds, cmps, hops = oracle.search(b, "ip", z["offsets"], z["nbrs"], int(z["ep"]), q, k, L, nthreads=2)
        cosine_close(ids, ds, cmps, hops, z, tag)


def test_format_goldens(oracle, tmp_path):
    z = np.load(os.path.join(GOLD, "formats.npz"))
    p = str(tmp_path / "f.bin")
    for key in z.files:
        if not key.endswith("_says"):
            continue
        case = key[:-5]
        open(p, "wb").write(z[case].tobytes())
        said = str(z[key])
        fn = oracle.gt_meta if case.startswith("gt_") else oracle.fbin_meta
        if said.startswith("OK"):
            assert list(fn(p)) == [int(x) for x in said.split()[1:]], case
        else:
            with pytest.raises(RuntimeError, match="Data file size wrong!"):
                fn(p)
    open(p, "wb").write(z["gt_good"].tobytes())
    ids, ds = oracle.gt_load(p)
    assert (ids == z["gt_good_ids"]).all() and (bits(ds) == z["gt_good_dist_bits"]).all()
    open(p, "wb").write(z["fbin_good"].tobytes())
    arr, d = oracle.fbin_load(p)
    assert list(arr.shape) == list(z["fbin_good_loaded_shape"]) and (bits(arr) == z["fbin_good_loaded_bits"]).all()


def test_recall_definition(oracle):
    res = np.array([[1, 2, 3], [7, 8, 9]], np.uint32)
    gt = np.array([[3, 1, 5, 2], [9, 9, 1, 7]], np.uint32)   # only the first k columns count; repeats in gt count twice
    assert oracle.recall(res, gt, 3) == pytest.approx((2 + 2) / 6)


def test_search_edge_cases(oracle):
    """k > reachable nodes -> the reference's 'not enough results'; isolated entry point; L_pq = 1."""
    base = np.random.default_rng(0).standard_normal((50, 16)).astype(np.float32)
    q = base[:3] + 0.1
    off = np.zeros(51, np.uint64)           # no edges at all
    with pytest.raises(RuntimeError, match="not enough results"):
        oracle.search(base, "l2", off, np.zeros(0, np.uint32), 7, q, 2, 10)
    ids, ds, cmps, hops = oracle.search(base, "l2", off, np.zeros(0, np.uint32), 7, q, 1, 10)
    assert (ids == 7).all() and (cmps == 0).all() and (hops == 1).all()


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_blocked_sgemm_groundtruth_matches_fp64(oracle, metric):
    """oracle/gt_numpy.py (the CPU baseline of the bench's ground-truth leg: blocked SGEMM + per-block top-K) against the
    fp64 brute force: same ids up to fp32 near-ties, distances within 1e-4 relative."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import gt_numpy
    rng = np.random.default_rng(5)
    base = rng.standard_normal((6000, 40), dtype=np.float32)
    q = (0.3 + 0.5 * rng.standard_normal((32, 40), dtype=np.float32)).astype(np.float32)
    ids, ds = gt_numpy.groundtruth_blocked(base, q, metric, 50, block=1024)
    i64, d64, _ = oracle.groundtruth_f64(base, q, metric, 50)
    assert (ids == i64).mean() > 0.99
    assert np.allclose(np.sort(ds, axis=1), np.sort(d64, axis=1), rtol=1e-4, atol=1e-4)
