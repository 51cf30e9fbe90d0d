"""-m gpu: the opt-in bf16 fast mode (rg_index_set "fast_bf16").  It is NOT a parity mode, so the checks
are properties: the distances returned are the exact fp32 compare() values of the returned ids, the list is ordered by
(distance, id) without repeats, runs are deterministic, the overlap with the exact search is high, and the knob changes
nothing where the mode does not apply."""
import numpy as np
import pytest

from helpers import bits, small_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rg():
    from roargraph_amd import index
    from roargraph_amd._lib import lib
    assert lib().rg_device_count() >= 1, "no GPU visible: the HIP path cannot run and there is no fallback"
    return index


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000), ("l2", 200, 3000), ("ip", 512, 1500)])
@pytest.mark.parametrize("L,k", [(100, 10), (500, 100), (64, 1)])
@pytest.mark.parametrize("visited", [2, 0])
def test_fast_mode_properties(rg, oracle, metric, d, nb, L, k, visited):
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    exact = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
    ix.set("visited", visited)           # 2: LDS filter only under the fast mode, 0: exact HBM words
    ix.set("fast_bf16", 1)
    ids, dists, cmps, hops = ix.SearchRoarGraph(q, k, L)
    ids2, dists2, _, _ = ix.SearchRoarGraph(q, k, L)
    assert (ids == ids2).all() and (bits(dists) == bits(dists2)).all(), "not deterministic"
    overlap = 0
    for i in range(q.shape[0]):
        assert len(set(ids[i].tolist())) == k, "repeated id"
        want = oracle.score_batch(base, metric, q[i], ids[i])   # the checker's compare(), not the HIP operator
        assert (bits(dists[i]) == bits(want)).all(), "returned distances are not the exact fp32 distances of the ids"
        key = list(zip(dists[i].tolist(), ids[i].tolist()))
        assert key == sorted(key), "not ordered by (distance, id)"
        overlap += len(set(ids[i].tolist()) & set(exact[0][i].tolist()))
    assert overlap / (q.shape[0] * k) >= 0.9, "fast mode lost more than 10 % of the exact search's neighbours"
    assert (cmps > 0).all() and (hops > 0).all()
    ix.set("fast_bf16", 0)                                   # switching it off restores parity
    ix.set("visited", 2)
    back = ix.SearchRoarGraph(q, k, L)
    assert (back[0] == exact[0]).all() and (bits(back[1]) == bits(exact[1])).all() and (back[2] == exact[2]).all()
    ix.close()


def test_fast_mode_is_a_no_op_where_it_does_not_apply(rg, oracle):
    """Other dimensions than 200 / 512 have no bf16 instantiation: the knob must leave the result bit-exact."""
    base, q, off, nbrs, ep = small_set("l2", 3000, 24)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric="l2")
    ix.set("fast_bf16", 1)
    got = ix.SearchRoarGraph(q, 10, 100)
    want = oracle.search(base, "l2", off, nbrs, ep, q, 10, 100, nthreads=4)
    assert (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all() and (got[2] == want[2]).all()
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000), ("l2", 24, 3000)])
@pytest.mark.parametrize("L,k", [(100, 10), (500, 100), (16, 1)])
@pytest.mark.parametrize("visited", [2, 1, 0])
def test_multi_expand_properties(rg, oracle, metric, d, nb, L, k, visited):
    """rg_index_set "multi_expand" (opt-in, NOT parity): two expansions per iteration.  The traversal
    differs from the reference's, so the checks are properties: exact distances of the returned ids, (distance, id)
    order without repeats, determinism, every query expands at least as many nodes as its beam is wide, high overlap
    with the exact search, and the knob off restores parity bit for bit."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    exact = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
    ix.set("visited", visited)
    ix.set("multi_expand", 1)
    ids, dists, cmps, hops = ix.SearchRoarGraph(q, k, L)
    ids2, dists2, cmps2, hops2 = ix.SearchRoarGraph(q, k, L)
    assert (ids == ids2).all() and (bits(dists) == bits(dists2)).all() and (hops == hops2).all(), "not deterministic"
    if visited != 1:
        assert (cmps == cmps2).all()
    overlap = 0
    for i in range(q.shape[0]):
        assert len(set(ids[i].tolist())) == k, "repeated id"
        want = oracle.score_batch(base, metric, q[i], ids[i])   # the checker's compare(), not the HIP operator
        assert (bits(dists[i]) == bits(want)).all(), "returned distances are not the exact fp32 distances of the ids"
        key = list(zip(dists[i].tolist(), ids[i].tolist()))
        assert key == sorted(key), "not ordered by (distance, id)"
        overlap += len(set(ids[i].tolist()) & set(exact[0][i].tolist()))
    assert overlap / (q.shape[0] * k) >= 0.9
    assert (hops >= np.minimum(L, exact[3])).all() and (cmps > 0).all()
    ix.set("multi_expand", 0)
    ix.set("visited", 2)
    back = ix.SearchRoarGraph(q, k, L)
    assert (back[0] == exact[0]).all() and (bits(back[1]) == bits(exact[1])).all() and (back[2] == exact[2]).all() and (back[3] == exact[3]).all()
    ix.close()
