"""CPU, build container only: the C oracle against oracle/_ref/rg_ref, the driver compiled from the reference's own
headers (distance.h, neighbor.h, visited_list_pool.h, util.h) -- live, on fresh random inputs, beyond the goldens."""
import numpy as np
import pytest

from helpers import bits, small_set
from roargraph_amd import io

from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref/rg_ref not built (needs /root/reference) or no AVX-512")


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_distance_all_tail_shapes(metric):
    rng = np.random.default_rng(11)
    for d in list(range(1, 41)) + [63, 64, 65, 100, 128, 200, 203, 256, 512, 960]:
        a = rng.standard_normal((300, d)).astype(np.float32)
        b = (0.3 + 0.5 * rng.standard_normal((300, d))).astype(np.float32)
        want = po.ref_dist(metric, a, b)
        assert (bits(po.compare_pairs(metric, a, b)) == bits(want)).all(), d
        po.use_avx512(True)
        got = po.compare_pairs(metric, a, b)
        po.use_avx512(False)
        assert (bits(got) == bits(want)).all(), d


def test_queue_random_traces():
    rng = np.random.default_rng(5)
    for cap in (1, 2, 3, 7, 32, 200):
        n = 4000
        ops = (rng.random(n) < 0.3).astype(np.uint8)
        ids = rng.integers(0, 300, n).astype(np.uint32)
        ds = (rng.integers(0, 40, n) / 4.0).astype(np.float32)
        a, b = po.queue_trace(cap, ops, ids, ds), po.ref_queue(cap, ops, ids, ds)
        assert a["size"] == b["size"] and a["cur"] == b["cur"]
        assert (a["ids"] == b["ids"]).all() and (bits(a["dists"]) == bits(b["dists"])).all()
        assert (a["flags"] == b["flags"]).all() and (a["pops"][: len(b["pops"])] == b["pops"]).all()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 3000), ("l2", 136, 1500)])
def test_search_live(tmp_path, metric, d, nb):
    base, q, off, nbrs, ep = small_set(metric, nb, d, nq=40)
    bf, qf, gf = (str(tmp_path / x) for x in ("b.fbin", "q.fbin", "g.index"))
    io.write_fbin(bf, base); io.write_fbin(qf, q); io.write_index(gf, off, nbrs, ep)
    for L, k in ((10, 10), (77, 5), (300, 100)):
        r = po.ref_search(bf, gf, qf, metric, k, L, threads=2)
        o = po.search(base, metric, off, nbrs, ep, q, k, L, nthreads=2)
        assert (r[0] == o[0]).all() and (bits(r[1]) == bits(o[1])).all()
        assert (r[2] == o[2]).all() and (r[3] == o[3]).all()


def test_prefetching_loop_returns_the_same_bits(tmp_path):
    """rg_ref search with the reference's software prefetches (index_bipartite.cpp:2324, 2374-2375; the form bench.py times
    as cpu_baseline) and without them: prefetches change the time, never a result."""
    base, q, off, nbrs, ep = small_set("ip", 3000, 200, nq=40)
    bf, qf, gf = (str(tmp_path / x) for x in ("b.fbin", "q.fbin", "g.index"))
    io.write_fbin(bf, base); io.write_fbin(qf, q); io.write_index(gf, off, nbrs, ep)
    a = po.ref_search(bf, gf, qf, "ip", 10, 120, threads=2, prefetch=True)
    b = po.ref_search(bf, gf, qf, "ip", 10, 120, threads=2, prefetch=False)
    o = po.search(base, "ip", off, nbrs, ep, q, 10, 120, nthreads=2)
    for x, y in ((a, b), (a, o)):
        assert (x[0] == y[0]).all() and (bits(x[1]) == bits(y[1])).all() and (x[2] == y[2]).all() and (x[3] == y[3]).all()


@pytest.mark.parametrize("nd,d,offset", [(1, 8, 0.0), (300, 200, 0.0), (4097, 200, 100.0), (2500, 512, 0.0), (20000, 24, 3.0), (513, 104, 0.0)])
def test_projection_ep_restatement_vs_reference_flags(nd, d, offset):
    """CalculateProjectionep (src/index_bipartite.cpp:2004-2041): the oracle's index-order loops against the same loops
    compiled with the reference's -Ofast (rg_ref ep), where the compiler may re-associate the j sum.  The rows here are in
    general position (no exact ties), so both must name the same row."""
    rng = np.random.default_rng(nd * 7 + d)
    base = (rng.standard_normal((nd, d)) * 3 + offset).astype(np.float32)
    assert po.projection_ep(base) == po.ref_projection_ep(base)


def test_not_enough_results_message(tmp_path):
    base = np.random.default_rng(0).standard_normal((50, 16)).astype(np.float32)
    bf, qf, gf = (str(tmp_path / x) for x in ("b.fbin", "q.fbin", "g.index"))
    io.write_fbin(bf, base); io.write_fbin(qf, base[:2])
    io.write_index(gf, np.zeros(51, np.uint64), np.zeros(0, np.uint32), 3)
    with pytest.raises(RuntimeError, match="not enough results: 1, expected: 2"):
        po.ref_search(bf, gf, qf, "l2", 2, 10)
