"""-m gpu: the HIP path against the committed golden vectors (values the reference's own code produced here;
tests/golden/, scripts/make_golden.py).  Complements test_gpu_parity.py (HIP vs oracle on fresh inputs)."""
import glob
import os

import numpy as np
import pytest

from helpers import bits

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "dist_*.npz"))))
def test_distance_goldens_on_gpu(path):
    """Each golden pair (a_i, b_i): base row a_i scored against query b_i."""
    from roargraph_amd.index import IndexBipartite
    z = np.load(path)
    metric = os.path.basename(path).split("_")[1]
    a, b = z["a"], z["b"]
    n, d = a.shape
    if d % 8:
        pytest.skip("dims are padded to a multiple of 8 at load (util.h:37-75)")
    ix = IndexBipartite.from_arrays(a, np.zeros(n + 1, np.uint64), np.zeros(0, np.uint32), 0, metric=metric)
    for i in range(0, n, 7):
        got = ix.score_batch(b[i], np.array([i], np.uint32))
        assert bits(got)[0] == z["expect_bits"][i], (path, i)
    # and all rows against one query, batched
    got = ix.score_batch(b[1], np.arange(n, dtype=np.uint32))
    assert bits(got)[1] == z["expect_bits"][1]
    ix.close()


@pytest.mark.parametrize("name", ["ip200", "l2_512", "ip24"])
@pytest.mark.parametrize("layout", ["ell", "csr"])
def test_search_goldens_on_gpu(name, layout, monkeypatch):
    from roargraph_amd.index import IndexBipartite
    monkeypatch.setenv("RG_FORCE_CSR", "1" if layout == "csr" else "0")
    z = np.load(os.path.join(GOLD, "search_%s.npz" % name))
    metric = str(z["metric"])
    ix = IndexBipartite.from_arrays(z["base"], z["offsets"], z["nbrs"], int(z["ep"]), metric=metric)
    for tag in z["configs"]:
        L, k = (int(x[1:]) for x in str(tag).split("_"))
        for rpp in (4, 8, 16):
            ix.set("rows_per_pass", rpp)
            ix.set("visited", 1)    # LDS filter mode: everything but cmps must still match the reference
            ids, ds, cmps, hops = ix.SearchRoarGraph(z["queries"], k, L)
            assert (hops == z[tag + "_hops"]).all() and (ids == z[tag + "_ids"]).all(), (tag, rpp, "filter")
            assert (bits(ds) == z[tag + "_dist_bits"]).all() and (cmps >= z[tag + "_cmps"]).all(), (tag, rpp, "filter")
            ix.set("visited", 0)
            ids, ds, cmps, hops = ix.SearchRoarGraph(z["queries"], k, L)
            assert (cmps == z[tag + "_cmps"]).all() and (hops == z[tag + "_hops"]).all(), (tag, rpp)
            assert (ids == z[tag + "_ids"]).all(), (tag, rpp)
            assert (bits(ds) == z[tag + "_dist_bits"]).all(), (tag, rpp)
    ix.close()


@pytest.mark.parametrize("layout", ["ell", "csr"])
def test_search_golden_cosine_on_gpu(layout, monkeypatch, oracle):
    """a3 on the HIP path: rg_index_open_mem(metric = RG_METRIC_COSINE) normalises the base rows, rg_search the queries
    (test_search_roargraph.cpp:141-153, 167-172; index_bipartite.cpp:2679-2684), then the IP kernel.  Against the golden of
    the reference build (tolerance rule: tests/test_oracle_golden.py::cosine_close) and, bit for bit, against the oracle."""
    from test_oracle_golden import cosine_close
    from roargraph_amd.index import IndexBipartite
    monkeypatch.setenv("RG_FORCE_CSR", "1" if layout == "csr" else "0")
    z = np.load(os.path.join(GOLD, "search_cos200.npz"))
    ix = IndexBipartite.from_arrays(z["base"], z["offsets"], z["nbrs"], int(z["ep"]), metric="cosine")
    b, q = z["base"].copy(), z["queries"].copy()
    oracle.normalize_rows(b)
    oracle.normalize_rows(q)
    for tag in z["configs"]:
        L, k = (int(x[1:]) for x in str(tag).split("_"))
        for visited in (2, 0):
            ix.set("visited", visited)
            ids, ds, cmps, hops = ix.SearchRoarGraph(z["queries"], k, L)
            cosine_close(ids, ds, cmps, hops, z, tag)
            want = oracle.search(b, "ip", z["offsets"], z["nbrs"], int(z["ep"]), q, k, L, nthreads=2)
            assert (ids == want[0]).all() and (bits(ds) == bits(want[1])).all() and (cmps == want[2]).all() and (hops == want[3]).all()
    ix.close()


def test_not_enough_results_and_arg_errors():
    from roargraph_amd._lib import RG_ERR_ARG, RG_ERR_NOT_ENOUGH, RgError
    from roargraph_amd.index import IndexBipartite
    base = np.random.default_rng(0).standard_normal((50, 16)).astype(np.float32)
    ix = IndexBipartite.from_arrays(base, np.zeros(51, np.uint64), np.zeros(0, np.uint32), 3, metric="l2")
    with pytest.raises(RgError, match="not enough results: 1, expected: 2") as e:
        ix.SearchRoarGraph(base[:4], 2, 10)
    assert e.value.code == RG_ERR_NOT_ENOUGH
    ids, ds, cmps, hops = ix.SearchRoarGraph(base[:4], 1, 10)
    assert (ids == 3).all() and (cmps == 0).all() and (hops == 1).all()
    with pytest.raises(RgError, match="L_pq must greater or equal than k") as e:
        ix.SearchRoarGraph(base[:4], 11, 10)
    assert e.value.code == RG_ERR_ARG
    ix.close()
    bad = np.array([0, 1], np.uint64)
    with pytest.raises(RgError, match="node id >= npts"):
        IndexBipartite.from_arrays(base[:1], bad, np.array([5], np.uint32), 0)


def test_full_size_properties():
    """BASELINE-shaped property checks that do not need the oracle: determinism across launches and knob settings,
    sortedness of the returned (distance, id) pairs, ids in range, cmps >= hops-ish invariants, on a 2M x 200 base."""
    import torch
    from roargraph_amd.index import IndexBipartite
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    nb, deg, nq, k, L = 2_000_000, 32, 2000, 10, 200
    base = torch.empty((nb, 200), device=dev).normal_(generator=g)
    nbrs = torch.randint(0, nb, (nb * deg,), dtype=torch.int32, device=dev, generator=g)
    off = torch.arange(0, nb + 1, dtype=torch.int64, device=dev) * deg
    q = torch.empty((nq, 200), device=dev).normal_(generator=g)
    ix = IndexBipartite.from_device(base, off, nbrs, 0, metric="ip")
    outs = []
    for wpc, rpp in ((0, 8), (4, 4), (16, 16)):
        ix.set("waves_per_cu", wpc); ix.set("rows_per_pass", rpp)
        ids = torch.zeros((nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((nq, k), device=dev)
        cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
        ix.search_dev(q, k, L, ids, ds, cm, hp); ix.search_wait()
        outs.append((ids.cpu(), ds.cpu(), cm.cpu(), hp.cpu()))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0])), "results depend on launch geometry"
    ids, ds, cm, hp = outs[0]
    assert (ids >= 0).all() and (ids < nb).all()
    assert (ds[:, 1:] >= ds[:, :-1]).all(), "distances not ascending"
    assert (hp >= L).all() and (cm <= hp * deg).all() and (cm >= L - 1).all()
    # returned distances are the true scores of the returned ids
    chk = -(base[ids[:50].long().reshape(-1)] * q[:50].repeat_interleave(k, 0)).sum(1).cpu().reshape(50, k)
    assert torch.allclose(chk, ds[:50], rtol=1e-4, atol=1e-4)
    ix.close()
