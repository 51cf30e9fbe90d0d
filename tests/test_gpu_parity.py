"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: neighbour ids, cmps and hops bit-exact; distances bit-exact (north star allows 1e-4 relative, we hold 0 ulp).
"""
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


@pytest.mark.parametrize("metric,d", [("ip", 200), ("l2", 512), ("ip", 512), ("l2", 200), ("ip", 8), ("l2", 24),
                                      ("ip", 64), ("l2", 136), ("ip", 960)])
def test_score_batch_bit_exact(rg, oracle, metric, d):
    rng = np.random.default_rng(d)
    nb = 3000
    base = rng.standard_normal((nb, d)).astype(np.float32)
    q = (0.3 + 0.5 * rng.standard_normal(d)).astype(np.float32)
    off = np.zeros(nb + 1, np.uint64)
    ix = rg.IndexBipartite.from_arrays(base, off, np.zeros(0, np.uint32), 0, metric=metric)
    for n in (1, 3, 4, 5, 8, 9, 63, 64, 65, 1000, 4097):
        ids = rng.integers(0, nb, n).astype(np.uint32)
        got = ix.score_batch(q, ids)
        want = oracle.score_batch(base, metric, q, ids)
        assert (bits(got) == bits(want)).all(), (metric, d, n)
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000), ("l2", 200, 3000), ("ip", 512, 1500)])
@pytest.mark.parametrize("L,k", [(10, 10), (50, 10), (100, 100), (500, 10), (64, 1), (65, 65), (2000, 100)])
def test_search_bit_exact(rg, oracle, metric, d, nb, L, k):
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    got = ix.SearchRoarGraph(q, k, L)
    want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
    assert (got[2] == want[2]).all(), "cmps differ"
    assert (got[3] == want[3]).all(), "hops differ"
    assert (got[0] == want[0]).all(), "neighbour ids differ"
    assert (bits(got[1]) == bits(want[1])).all(), "distance bits differ"
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000)])
@pytest.mark.parametrize("L,k", [(10, 10), (100, 100), (500, 10)])
@pytest.mark.parametrize("flt", [4, 9, 11])
def test_search_lds_filter_mode(rg, oracle, metric, d, nb, L, k, flt):
    """visited=1: the LDS exact-match filter may forget a node and score it again, which cannot change the beam
    (same distance bits -> dropped as a repeat or rejected by the tail).  ids, distances and hops stay bit-exact;
    cmps counts the evaluations actually performed (>= the reference's)."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ix.set("visited", 1)
    ix.set("filter_log2", flt)
    got = ix.SearchRoarGraph(q, k, L)
    want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
    assert (got[3] == want[3]).all(), "hops differ"
    assert (got[0] == want[0]).all(), "neighbour ids differ"
    assert (bits(got[1]) == bits(want[1])).all(), "distance bits differ"
    assert (got[2] >= want[2]).all(), "fewer evaluations than the reference is impossible"
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 24, 3000), ("l2", 512, 2000), ("ip", 24, 70000)])
@pytest.mark.parametrize("wpc", [0, 13, 9, 7, 3])
def test_filter_of_any_size(rg, oracle, metric, d, nb, wpc):
    """filter_fill: the LDS visited filter takes the LDS the resident queries leave, so its slot count is whatever fits -- not a
    power of two.  slot = floor(hash * slots / 2^id_bits), entry = the low bits of the hash: still one to one with the id, so
    a hit still proves "visited" and every output stays the oracle's.  filter_fill = 2 fills on small batches too; the
    resident count (waves_per_cu) varies what is left over, i.e. the slot count; a 70,000-node index has 17-bit ids (entries
    of several bits over tables that are not a power of two).  In the look-ahead form of the exact words the same region is
    a one-hash bit screen (a clear bit proves "not visited", the word is then not read): same outputs again."""
    if nb > 10000:      # random 20-regular graph (repeats and self loops included)
        from roargraph_amd import synth
        base, q = synth.make_synth(5, nb, 40, d)
        off, nbrs = synth.random_regular_csr(nb, 20)
        ep = 17
    else:
        base, q, off, nbrs, ep = small_set(metric, nb, d, nq=100)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ix.set("filter_fill", 2)
    ix.set("waves_per_cu", wpc)
    for L, k in ((10, 10), (100, 100), (700, 10), (1900, 10)):
        want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
        for visited, look in ((2, -1), (1, -1), (0, 0), (0, 1)):   # look-ahead form: the region is the bit screen of the words
            ix.set("visited", visited)
            ix.set("lookahead", look)
            got = ix.SearchRoarGraph(q, k, L)
            assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, visited, look)
            assert (got[2] == want[2]).all() if visited != 1 else (got[2] >= want[2]).all(), ("cmps", L, visited, look)
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 24, 3000), ("l2", 512, 2000), ("ip", 136, 2500)])
@pytest.mark.parametrize("long_ep_row", [False, True])
def test_shared_frontier_mode_is_exact(rg, oracle, metric, d, nb, long_ep_row):
    """shared_frontier = 1 (third mode; opt-in): the rows every query of a batch scores first -- the entry point
    and its neighbours -- are scored once for the whole batch by rg_front_score_kernel with the exact routine, and K1's
    first hop reads those scores instead of gathering the rows.  Same bits, so every output stays the oracle's: all visited
    modes, an entry point with more than 63 neighbours (two chunks), sub-batches over a small log budget."""
    from roargraph_amd import io
    base, q, off, nbrs, ep = small_set(metric, nb, d, nq=90)
    if long_ep_row:
        lists = [nbrs[int(off[i]):int(off[i + 1])].copy() for i in range(nb)]
        lists[ep] = np.unique(np.concatenate([lists[ep], (np.arange(ep + 1, ep + 100) % nb).astype(np.uint32)])).astype(np.uint32)
        lists[ep] = lists[ep][lists[ep] != ep]
        off, nbrs = io.lists_to_csr(lists)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ix.set("shared_frontier", 1)
    for L, k in ((10, 10), (100, 100), (1300, 10)):
        want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
        for visited, budget in ((2, 0), (1, 0), (0, 0), (2, 600)):
            ix.set("visited", visited)
            ix.set("log_budget_kb", budget if budget else (16 << 20))
            ix.set("log_cap", 4096 if budget else 0)
            got = ix.SearchRoarGraph(q, k, L)
            assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, visited, budget)
            assert (got[2] == want[2]).all() if visited != 1 else (got[2] >= want[2]).all(), ("cmps", L, visited, budget)
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 24, 3000), ("l2", 512, 2000)])
@pytest.mark.parametrize("min_indeg", [2, 6, 40, 255])
def test_filter_admission_by_in_degree(rg, oracle, metric, d, nb, min_indeg):
    """filter_min_indeg: the LDS visited filter keeps entries only for neighbours whose in-degree (carried in the top byte of
    the adjacency words) reaches the threshold -- a node can be met again at most in-degree - 1 times.  Nodes below it are
    never remembered, i.e. possibly scored again, which cannot change the beam: ids / distances / hops and (through the
    id log + K4, or the exact words behind the filter) cmps stay bit-exact at any threshold."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ix.set("filter_min_indeg", min_indeg)
    for L, k in ((10, 10), (100, 100), (600, 10)):
        want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
        for visited in (2, 1, 0):
            ix.set("visited", visited)
            got = ix.SearchRoarGraph(q, k, L)
            assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, visited)
            assert (got[2] == want[2]).all() if visited != 1 else (got[2] >= want[2]).all(), ("cmps", L, visited)
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000), ("l2", 200, 3000), ("ip", 512, 1500)])
@pytest.mark.parametrize("visited", [2, 1, 0])
def test_compute_layout_gather_form(rg, oracle, metric, d, nb, visited):
    """gather_form = 1: rows fetched in the layout the FMAs consume (one dword per lane and step, no LDS bounce) instead of
    16 bytes per lane + transposition.  The accumulation order is the same, so every output bit is: all visited modes, with
    and without the split-row copy, every register-set depth that has the form."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ix.set("gather_form", 1)
    ix.set("visited", visited)
    for L, k in ((10, 10), (100, 100), (700, 10)):
        want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
        for rpp in ((16, 32) if d == 200 else (8,)):
            for split in ((1, 0) if d == 200 else (1,)):
                ix.set("rows_per_pass", rpp)
                ix.set("split_rows", split)
                got = ix.SearchRoarGraph(q, k, L)
                assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, rpp, split)
                assert (got[2] == want[2]).all() if visited != 1 else (got[2] >= want[2]).all(), ("cmps", L, rpp, split)
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000), ("l2", 200, 3000), ("ip", 512, 1500)])
@pytest.mark.parametrize("lookahead,exact_filter", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_exact_words_forms(rg, oracle, metric, d, nb, lookahead, exact_filter):
    """visited=0, the reference's tag array (visited_list_pool.h:8-29) as epoch-tagged words in HBM, in both kernel forms:
    look-ahead (plain-load test, fire-and-forget marks, the next pop's adjacency row and words fetched early) and
    returning atomics; with and without the LDS filter in front; every register-set depth.  All four outputs bit-exact."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ix.set("visited", 0)
    ix.set("lookahead", lookahead)
    ix.set("exact_filter", exact_filter)
    for L, k in ((10, 10), (100, 100), (700, 10), (2000, 10)):
        want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
        for rpp in ((8, 16, 32) if d == 200 else (8, 16)):
            ix.set("rows_per_pass", rpp)
            for vbytes in ((-1, 0) if lookahead else (-1,)):   # look-ahead form: one epoch byte per node (default) or the words
                ix.set("visited_bytes", vbytes)
                for rep in range(2):        # the second call re-uses the slots' visited words under the next epochs
                    got = ix.SearchRoarGraph(q, k, L)
                    assert (got[2] == want[2]).all(), ("cmps", L, rpp, vbytes, rep)
                    assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, rpp, vbytes, rep)
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000)])
@pytest.mark.parametrize("front_set", [-1, 30, 90])
def test_look_ahead_form_with_an_exact_set_in_front_of_the_tags(rg, oracle, metric, d, nb, front_set):
    """Knob front_set (round 4): in the look-ahead byte-tag form the front of the LDS region is an exact set (the buckets of the
    narrow-beam form); a node it holds needs no tag, the tags and the bit screen keep the nodes it had no room for.  Small and
    large shares of the region, beams that fit the set and beams that outgrow it many times, repeated calls (the slots' epochs
    advance), the wrap of the epoch byte with two slots: all four outputs bit-exact."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ix.set("visited", 0)
    ix.set("lookahead", 1)
    ix.set("front_set", front_set)
    for L, k in ((10, 10), (100, 100), (700, 10), (2000, 10)):
        want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
        for rpp in ((16, 32) if d == 200 else (8, 16)):
            ix.set("rows_per_pass", rpp)
            for rep in range(2):
                got = ix.SearchRoarGraph(q, k, L)
                assert (got[2] == want[2]).all(), ("cmps", L, rpp, rep)
                assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, rpp, rep)
    ix.set("visited_budget_kb", 8)      # two slots: both pass the wrap of the epoch byte below
    want = oracle.search(base, metric, off, nbrs, ep, q, 10, 300, nthreads=4)
    for call in range(10):
        got = ix.SearchRoarGraph(q, 10, 300)
        assert (got[2] == want[2]).all() and (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), call
    ix.close()


def test_byte_tags_survive_the_epoch_wrap(rg, oracle):
    """Byte form of the exact visited set (look-ahead kernel form): a slot's epoch is one byte, so its tags are wiped every 255
    queries (VisitedList::reset, visited_list_pool.h:20-26).  Two slots (visited_budget_kb) serve 64 queries per call: twelve
    calls take both slots through the wrap; every call must return the oracle's outputs."""
    base, q, off, nbrs, ep = small_set("ip", 4000, 200)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip")
    ix.set("visited", 0)
    ix.set("lookahead", 1)
    ix.set("visited_budget_kb", 8)      # 4000 tag bytes per slot (rounded to 4096): two slots
    want = oracle.search(base, "ip", off, nbrs, ep, q, 10, 300, nthreads=4)
    for call in range(12):
        got = ix.SearchRoarGraph(q, 10, 300)
        assert (got[2] == want[2]).all() and (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), call
    ix.close()


@pytest.mark.parametrize("lookahead", [1, 0])
def test_exact_words_on_rows_with_repeats_and_long_rows(rg, oracle, lookahead):
    """The file format allows a list to name a node twice (SURVEY C-10): the second occurrence must not be scored.  An
    index with such a list -- of ANY length: short, 64 .. 126 neighbours (the look-ahead form's two-read step), longer (its
    general path) -- is detected at open and keeps the returning-atomic form whatever the knob says; lists longer than one
    63-neighbour read take the general path inside the look-ahead form."""
    from roargraph_amd import io
    base, q, off, nbrs, ep = small_set("ip", 4000, 200)
    lists = [nbrs[int(off[i]):int(off[i + 1])].copy() for i in range(base.shape[0])]
    long_rows = dict((i, np.unique(np.concatenate([lists[i], np.arange(i + 1, i + 120, dtype=np.uint32) % 4000]))) for i in (ep, 17, 900))
    first = int(lists[ep][0])      # expanded by every query

    def mid_row(i):                # 100 distinct neighbours, the 3rd named again at position 90 (both in the second read)
        r = np.unique(np.concatenate([lists[i], np.arange(i + 1, i + 200, dtype=np.uint32) % 4000]))[:100].astype(np.uint32)
        return np.concatenate([r[:90], r[2:3], r[90:]])

    def very_long_row(i):          # 150 distinct neighbours, one named again beyond position 126
        r = np.unique(np.concatenate([lists[i], np.arange(i + 1, i + 300, dtype=np.uint32) % 4000]))[:150].astype(np.uint32)
        return np.concatenate([r[:140], r[70:71], r[140:]])

    for with_repeats in (None, "short", "mid", "long"):
        ls = list(lists)
        for i, row in long_rows.items():
            ls[i] = row.astype(np.uint32)
        if with_repeats == "short":
            ls[3] = np.concatenate([ls[3], ls[3][:2]]).astype(np.uint32)
            ls[first] = np.concatenate([ls[first][:5], ls[first][:5]]).astype(np.uint32)
        elif with_repeats == "mid":
            ls[first] = mid_row(first)
        elif with_repeats == "long":
            ls[first] = very_long_row(first)
        o2, n2 = io.lists_to_csr(ls)
        ix = rg.IndexBipartite.from_arrays(base, o2, n2, ep, metric="ip")
        ix.set("visited", 0)
        ix.set("lookahead", lookahead)
        for L, k in ((20, 10), (300, 10)):
            got = ix.SearchRoarGraph(q, k, L)
            want = oracle.search(base, "ip", o2, n2, ep, q, k, L, nthreads=4)
            assert (got[2] == want[2]).all(), ("cmps", with_repeats, L)
            assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all()
            assert all(len(set(r.tolist())) == k for r in got[0]), ("a result row names a node twice", with_repeats, L)
        ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 24, 3000), ("l2", 512, 2000)])
@pytest.mark.parametrize("count_in_k1", [-1, 0, 1000])
def test_distinct_count_inside_k1(rg, oracle, metric, d, nb, count_in_k1):
    """Default visited mode: beams up to "count_in_k1" wide (default 40) count the distinct ids of their log inside K1, at the
    end of each query (K4's bucket set in the LDS the beam and the filter no longer need, hash partitions when the log is
    longer than the table); wider beams -- or the knob at 0 -- leave the count to K4.  cmps is the reference's either way,
    with a forgetful filter that makes re-scored nodes plentiful."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ix.set("count_in_k1", count_in_k1)
    for flt in (0, 5):
        ix.set("filter_log2", flt)
        for L, k in ((10, 10), (40, 10), (41, 10), (150, 100), (700, 10)):
            got = ix.SearchRoarGraph(q, k, L)
            want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
            assert (got[2] == want[2]).all(), ("cmps", L, flt)
            assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, flt)
    ix.close()


@pytest.mark.parametrize("full_ids", [0, 1])
@pytest.mark.parametrize("log_cap,table", [(0, 15), (64, 15), (100000, 7), (700, 8), (100000, 6), (100000, 11)])
def test_default_mode_exact_cmps_paths(rg, oracle, log_cap, table, full_ids):
    """visited=2 (default): LDS filter + id log + exact distinct count.  Forced corner paths: logs that overflow
    (exact fallback pass re-counts those queries), tiny K4 tables (multi-partition counting, full buckets spilling into
    the side table) and both K4 set layouts (16-bit remainders / full ids)."""
    base, q, off, nbrs, ep = small_set("ip", 4000, 200)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip")
    ix.set("filter_log2", 6)            # forgetful filter -> plenty of re-scored nodes to de-duplicate
    ix.set("log_cap", log_cap)
    ix.set("count_table_log2", table)
    ix.set("count_full_ids", full_ids)
    for L, k in ((10, 10), (200, 10)):
        got = ix.SearchRoarGraph(q, k, L)
        want = oracle.search(base, "ip", off, nbrs, ep, q, k, L, nthreads=4)
        assert (got[2] == want[2]).all(), "cmps differ"
        assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all()
    ix.close()


@pytest.mark.parametrize("log_cap,budget_kb", [(65536, 1024), (8192, 1024), (300, 64)])
def test_large_batch_is_searched_in_sub_batches(rg, oracle, log_cap, budget_kb):
    """A batch whose id logs exceed the log budget runs as sub-batches over the same logs (4, 32 or 54 queries each
    here): every output still equals the oracle's -- also when logs overflow into the exact fallback pass (cap 300)."""
    base, q, off, nbrs, ep = small_set("ip", 4000, 200, nq=300)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip")
    ix.set("filter_log2", 6)
    ix.set("log_cap", log_cap)
    ix.set("log_budget_kb", budget_kb)   # budget / (log_cap * 4 B) queries per sub-batch
    got = ix.SearchRoarGraph(q, 10, 100)
    want = oracle.search(base, "ip", off, nbrs, ep, q, 10, 100, nthreads=4)
    assert (got[2] == want[2]).all(), "cmps differ"
    assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all()
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000)])
def test_register_and_lds_query_forms_agree(rg, oracle, metric, d, nb):
    """d = 200 / 512 run the K1 instantiation that keeps the query in registers; "query_in_lds" forces the generic
    one.  Both must equal the oracle bit for bit."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    want = oracle.search(base, metric, off, nbrs, ep, q, 10, 100, nthreads=4)
    for in_lds in (0, 1):
        ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
        ix.set("query_in_lds", in_lds)
        got = ix.SearchRoarGraph(q, 10, 100)
        assert (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all()
        assert (got[2] == want[2]).all() and (got[3] == want[3]).all()
        ix.close()


@pytest.mark.parametrize("nrep", [1, 2, 3])
def test_query_sharded_search_over_replicas(rg, oracle, nrep):
    """rg_search_sharded: replicas of the index (here all on device 0, which exercises the same code as one per GPU)
    search contiguous slices of the batch concurrently; the assembled result equals the oracle's, including a batch
    that does not divide evenly and a deferred not-enough-results report that names the query in the whole batch."""
    base, q, off, nbrs, ep = small_set("ip", 4000, 200, nq=101)
    reps = [rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip") for _ in range(nrep)]
    got = rg.search_sharded(reps, q, 10, 100)
    want = oracle.search(base, "ip", off, nbrs, ep, q, 10, 100, nthreads=4)
    assert (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all()
    assert (got[2] == want[2]).all() and (got[3] == want[3]).all()
    with pytest.raises(Exception, match="L_pq must greater or equal than k"):
        rg.search_sharded(reps, q, 20, 10)
    for r in reps:
        r.close()


def test_default_mode_stays_exact_when_it_switches_to_the_exact_words(rg, oracle):
    """The default visited mode is adaptive: after a batch (of >= 1000 queries) in which the LDS filter re-scored more than
    30 % extra nodes, the next batch at that beam width runs on the exact HBM words as a timed trial and the faster
    form is kept.  Every call -- before, during and after the trial -- must return the oracle's ids / distances /
    cmps / hops."""
    base, q, off, nbrs, ep = small_set("ip", 4000, 200, nq=1200)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip")
    for L in (500, 500, 500, 1000, 1000, 100, 500):
        got = ix.SearchRoarGraph(q, 10, L)
        want = oracle.search(base, "ip", off, nbrs, ep, q, 10, L, nthreads=4)
        assert (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all()
        assert (got[2] == want[2]).all() and (got[3] == want[3]).all()
    ix.close()


def test_empty_batch_and_single_query(rg, oracle):
    """An empty query batch is a no-op (host and device forms); a batch of one query is searched like any other."""
    import torch
    base, q, off, nbrs, ep = small_set("ip", 4000, 200)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip")
    got = ix.SearchRoarGraph(np.zeros((0, 200), np.float32), 10, 50)
    assert got[0].shape == (0, 10) and got[2].shape == (0,)
    dev = torch.device("cuda", 0)
    e = torch.zeros((0, 200), device=dev)
    ix.search_dev(e, 10, 50, torch.zeros((0, 10), dtype=torch.int32, device=dev), torch.zeros((0, 10), device=dev),
                  torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev))
    ix.search_wait()
    got = ix.SearchRoarGraph(q[:1], 10, 50)
    want = oracle.search(base, "ip", off, nbrs, ep, q[:1], 10, 50)
    assert (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all() and (got[2] == want[2]).all()
    ix.close()


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_split_rows_equal_plain_rows(rg, oracle, metric, monkeypatch):
    """d = 200 with ELL adjacency searches a split copy of the base by default (first 192 elements of a row at a whole-line
    stride, the 8-element tail once per edge in adjacency order).  Same values, same order of operations: the outputs must
    be the oracle's with the copy, with the knob off, and with the copy never built (RG_SPLIT_ROWS=0) -- on a graph with
    degrees 0 ... 150 (tails beyond the first 64-word adjacency read), repeated edges, and in all three visited modes."""
    rng = np.random.default_rng(23)
    nb, d = 3000, 200
    base = rng.standard_normal((nb, d)).astype(np.float32)
    q = (rng.standard_normal((64, d)) * 0.5 + 0.3).astype(np.float32)
    deg = rng.integers(0, 151, nb)
    deg[0] = 150
    off = np.zeros(nb + 1, np.uint64); off[1:] = np.cumsum(deg)
    nbrs = np.concatenate([rng.choice(nb, int(k), replace=True) for k in deg]).astype(np.uint32)   # repeats inside a row
    want = {L: oracle.search(base, metric, off, nbrs, 0, q, 10, L, nthreads=4) for L in (20, 500)}

    def check(ix, what):
        for vis in (2, 0, 1):
            ix.set("visited", vis)
            for L, w in want.items():
                got = ix.SearchRoarGraph(q, 10, L)
                assert (got[0] == w[0]).all() and (bits(got[1]) == bits(w[1])).all(), (what, vis, L)
                assert (got[3] == w[3]).all(), (what, vis, L)
                if vis != 1:
                    assert (got[2] == w[2]).all(), (what, vis, L)

    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, 0, metric=metric)
    check(ix, "split")
    ix.set("split_rows", 0)
    check(ix, "knob off")
    ix.set("split_rows", 1)
    ix.set("multi_expand", 1)                   # opt-in mode over the split copy: still a valid search
    got = ix.SearchRoarGraph(q, 10, 500)
    assert (np.sort(got[1], axis=1) == got[1]).all() and (got[0] < nb).all()
    ix.close()
    monkeypatch.setenv("RG_SPLIT_ROWS", "0")
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, 0, metric=metric)
    check(ix, "never built")
    ix.close()


@pytest.mark.parametrize("layout", ["ell", "csr"])
def test_wide_and_empty_adjacency_rows(rg, oracle, layout, monkeypatch):
    """Adjacency rows wider than one 64-word read (degrees up to 150) and rows of degree 0, both layouts."""
    monkeypatch.setenv("RG_FORCE_CSR", "1" if layout == "csr" else "0")
    rng = np.random.default_rng(17)
    nb, d = 3000, 200
    base = rng.standard_normal((nb, d)).astype(np.float32)
    q = (rng.standard_normal((48, d)) * 0.5 + 0.3).astype(np.float32)
    deg = rng.integers(0, 151, nb)
    deg[rng.integers(0, nb, 200)] = 0
    deg[0] = 150                                        # the entry point is well connected
    off = np.zeros(nb + 1, np.uint64); off[1:] = np.cumsum(deg)
    nbrs = np.concatenate([rng.choice(nb, int(k), replace=False) for k in deg]).astype(np.uint32)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, 0, metric="ip")
    for L in (50, 500):
        got = ix.SearchRoarGraph(q, 10, L)
        want = oracle.search(base, "ip", off, nbrs, 0, q, 10, L, nthreads=4)
        assert (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all()
        assert (got[2] == want[2]).all() and (got[3] == want[3]).all()
    ix.close()


@pytest.mark.parametrize("metric,d", [("ip", 200), ("l2", 200)])
@pytest.mark.parametrize("gather_form", [1, 0])
def test_rolled_gather_on_wide_hops(rg, oracle, metric, d, gather_form):
    """gather_roll = 1 (eight register sets): a hop with more than 32 fresh rows fetches its second batch of rows set by set
    behind the first instead of after it.  Rows of up to 126 distinct neighbours give hops of one to four batches, with every
    remainder; all visited forms, both gather layouts, split rows on and off: every output is the oracle's."""
    rng = np.random.default_rng(23)
    nb = 3000
    base = rng.standard_normal((nb, d)).astype(np.float32)
    q = (rng.standard_normal((40, d)) * 0.5 + 0.3).astype(np.float32)
    deg = rng.integers(1, 127, nb)
    deg[0] = 126
    off = np.zeros(nb + 1, np.uint64); off[1:] = np.cumsum(deg)
    nbrs = np.concatenate([rng.choice(nb, int(k), replace=False) for k in deg]).astype(np.uint32)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, 0, metric=metric)
    ix.set("gather_roll", 1)
    ix.set("rows_per_pass", 32)
    ix.set("gather_form", gather_form)
    for L, k in ((20, 10), (300, 100)):
        want = oracle.search(base, metric, off, nbrs, 0, q, k, L, nthreads=4)
        for visited, look, split in ((2, -1, 1), (1, -1, 0), (0, 1, 1), (0, 0, 1), (0, 1, 0)):
            ix.set("visited", visited); ix.set("lookahead", look); ix.set("split_rows", split)
            got = ix.SearchRoarGraph(q, k, L)
            assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, visited, look, split)
            assert (got[2] == want[2]).all() if visited != 1 else (got[2] >= want[2]).all(), ("cmps", L, visited, look, split)
    ix.close()


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_ties_duplicate_edges_and_self_loops(rg, oracle, metric):
    """Exact distance ties (every base row appears four times), repeated edges, self-loops and a single-query batch: the
    (distance, id) order and the visited rules must still match the oracle bit for bit, with k == L_pq too."""
    rng = np.random.default_rng(23)
    uniq, d = 700, 200
    base = np.tile(rng.standard_normal((uniq, d)).astype(np.float32), (4, 1))
    nb = base.shape[0]
    q = (rng.standard_normal((33, d)) * 0.5 + 0.3).astype(np.float32)
    lists = []
    for i in range(nb):
        a = rng.integers(0, nb, 20)
        a[:3] = a[3:6]                                   # repeated edges
        a[6] = i                                         # self-loop
        lists.append(a.astype(np.uint32))
    off = np.zeros(nb + 1, np.uint64); off[1:] = np.cumsum([len(x) for x in lists])
    nbrs = np.concatenate(lists)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, 5, metric=metric)
    for vis in (2, 1, 0):
        ix.set("visited", vis)
        for qq, k, L in ((q, 10, 100), (q[:1], 1, 1), (q, 64, 64), (q[:2], 100, 300)):
            got = ix.SearchRoarGraph(qq, k, L)
            want = oracle.search(base, metric, off, nbrs, 5, qq, k, L, nthreads=4)
            assert (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all() and (got[3] == want[3]).all()
            if vis != 1:
                assert (got[2] == want[2]).all()
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000)])
@pytest.mark.parametrize("lset_bytes", [0, 2048, 256])
def test_exact_set_in_lds(rg, oracle, metric, d, nb, lset_bytes):
    """Default visited mode at narrow beams (round 4): the exact visited set in LDS (K1 VIS = 3) -- no id log, no K4, no
    de-duplicating inserts; cmps exact as counted.  lset_bytes caps the set so that queries outgrow it: they finish in the
    forgetful form and count their own log (2048 bytes: some queries; 256: every query, at once)."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ix.set("lset", 100000)          # the form at every beam width, whatever the estimate says
    ix.set("lset_bytes", lset_bytes)
    for L, k in ((10, 10), (50, 10), (150, 100), (700, 10)):
        for rep in range(2):
            got = ix.SearchRoarGraph(q, k, L)
            want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
            assert (got[2] == want[2]).all(), ("cmps", L, lset_bytes, rep)
            assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, lset_bytes)
    assert ix.stat("batches_lset") == 8 and ix.stat("batches_filter_log") == 0
    if lset_bytes:
        assert ix.stat("lset_left") > 0
    # the automatic rule: the form where the estimate says the visits fit; a width at which queries outgrow the set is left
    ix.set("lset", -1)
    n0 = ix.stat("batches_lset")
    for rep in range(3):
        got = ix.SearchRoarGraph(q, 10, 20)
        want = oracle.search(base, metric, off, nbrs, ep, q, 10, 20, nthreads=4)
        assert (got[2] == want[2]).all() and (got[0] == want[0]).all()
    if lset_bytes == 0:
        assert ix.stat("batches_lset") == n0 + 3
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000)])
@pytest.mark.parametrize("lset_bytes", [2048, 256])
def test_exact_set_in_lds_with_tags_behind_it(rg, oracle, metric, d, nb, lset_bytes):
    """Knob lset_tags: the nodes the exact LDS set has no room for go to the exact epoch bytes in HBM (K1 VIS = 3 with
    P.visited): nothing is forgotten, cmps exact as counted -- at every width, over repeated launches
    (the epoch of a slot advances with every query it serves)."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ix.set("lset_tags", 2)          # wherever the set alone is not chosen, whatever it holds
    ix.set("lset_bytes", lset_bytes)
    for L, k in ((10, 10), (50, 10), (150, 100), (700, 10)):
        for rep in range(3):
            got = ix.SearchRoarGraph(q, k, L)
            want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
            assert (got[2] == want[2]).all(), ("cmps", L, lset_bytes, rep)
            assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, lset_bytes)
    assert ix.stat("batches_lset") == 12 and ix.stat("batches_filter_log") == 0 and ix.stat("batches_exact_hbm") == 0
    ix.close()


@pytest.mark.parametrize("metric,d,nb", [("ip", 200, 4000), ("l2", 512, 2000), ("ip", 24, 3000)])
@pytest.mark.parametrize("hub_bits,front_set", [(8, 0), (10, 0), (12, -1), (14, 40), (-1, -1), (0, 0)])
def test_hub_bits_in_front_of_the_tags(rg, oracle, metric, d, nb, hub_bits, front_set):
    """Round 5, knob hub_bits: the front of the look-ahead form's LDS region is an exact bitmap of 2^m bits, one bit per position
    of the hashed id, owned by the node of highest in-degree that falls on it (level stored in the top nibble of every ELL
    neighbour word at open).  A hub's bit is its visited flag -- no tag, no set entry, no screen bit.  Every bitmap size a launch
    can be given (2^8 bits: a handful of hubs; 2^14: most nodes of a small index are hubs), with and without the exact set
    behind it, long rows (the form's general path), repeated calls, the epoch wrap of the tags: all four outputs bit-exact."""
    base, q, off, nbrs, ep = small_set(metric, nb, d)
    from roargraph_amd import io
    lists = [nbrs[int(off[i]):int(off[i + 1])].copy() for i in range(base.shape[0])]
    for i in (ep, 17, 900):      # rows of 64 .. 126 and of more than 126 neighbours: the two-read step and the general path
        lists[i] = np.unique(np.concatenate([lists[i], np.arange(i + 1, i + (100 if i != 900 else 200), dtype=np.uint32) % nb])).astype(np.uint32)
    off, nbrs = io.lists_to_csr(lists)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    assert ix.stat("hub_levels") == 1
    ix.set("visited", 0)
    ix.set("lookahead", 1)
    ix.set("hub_bits", hub_bits)
    ix.set("front_set", front_set)
    dimc = d in (200, 512)       # the look-ahead form exists for the register-staged instantiations
    for L, k in ((10, 10), (100, 100), (700, 10), (2000, 10)):
        want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=4)
        for rpp in ((16, 32) if d == 200 else (8,)):
            ix.set("rows_per_pass", rpp)
            for rep in range(2):
                got = ix.SearchRoarGraph(q, k, L)
                assert (got[2] == want[2]).all(), ("cmps", L, rpp, rep)
                assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (L, rpp, rep)
            if dimc and hub_bits > 0:      # (a forced size the launch's region cannot hold -- 2^14 bits beside a 2000-entry beam -- is dropped)
                assert ix.stat("hub_m_last") in (0, hub_bits) and (L > 700 or ix.stat("hub_m_last") == hub_bits), (L, rpp)
            if hub_bits == 0 or not dimc:
                assert ix.stat("hub_m_last") == 0
    if dimc:
        ix.set("visited_budget_kb", 8)      # two slots: both pass the wrap of the epoch byte below
        want = oracle.search(base, metric, off, nbrs, ep, q, 10, 300, nthreads=4)
        for call in range(10):
            got = ix.SearchRoarGraph(q, 10, 300)
            assert (got[2] == want[2]).all() and (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), call
    ix.close()


def test_hub_levels_are_the_per_position_maxima_of_in_degree(rg, oracle):
    """The hub level written at open (top nibble of the ELL neighbour words) against a numpy restatement of its definition: at
    2^m bits, position p belongs to the node of largest (in-degree, -id) among those with (id * 0x9E3779B1) >> (32 - m) == p;
    level = the smallest m in 8 .. 22 at which a node owns its position, minus 8 (15 = never)."""
    import ctypes as C
    from roargraph_amd._lib import check, lib
    base, q, off, nbrs, ep = small_set("ip", 4000, 200)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip")
    nd = base.shape[0]
    indeg = np.bincount(nbrs, minlength=nd).astype(np.int64)
    x = (np.arange(nd, dtype=np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(0xffffffff)
    key = (indeg << 32) | (0xffffffff - np.arange(nd, dtype=np.int64))
    lvl = np.full(nd, 15, np.int64)
    for m in range(22, 7, -1):
        pos = (x >> np.uint64(32 - m)).astype(np.int64)
        best = np.zeros(1 << m, np.int64)
        np.maximum.at(best, pos, np.where(indeg > 0, key, 0))
        own = (indeg > 0) & (best[pos] == key)
        lvl[own] = m - 8
    stride = C.c_uint32()
    n = C.c_uint64()
    check(lib().rg_index_debug_ell(ix.handle, None, C.byref(n), C.byref(stride)))
    ell = np.zeros(n.value, np.uint32)
    check(lib().rg_index_debug_ell(ix.handle, ell.ctypes.data_as(C.c_void_p), C.byref(n), C.byref(stride)))
    ell = ell.reshape(nd, stride.value)
    for v in range(0, nd, 7):
        dg = int(ell[v, 0])
        w = ell[v, 1:1 + dg]
        ids = w & 0xffffff
        assert (ids == nbrs[int(off[v]):int(off[v + 1])]).all()
        assert ((w >> 28) == lvl[ids]).all() and (((w >> 24) & 15) == np.minimum(15, indeg[ids])).all(), v
    ix.close()


@pytest.mark.parametrize("lset_bytes", [256, 2048])
def test_exact_set_with_tags_is_not_used_over_rows_with_repeats(rg, oracle, lset_bytes):
    """the exact LDS set with the byte tags behind it tests a hop's neighbours lane-parallel; two lanes that bring the
    same node and find no room in the set would both read a stale tag and both score it.  On an index whose rows name a node
    twice the form must not be chosen (the pure set settles such lanes with its CAS, the logging form de-duplicates): forced
    with lset_tags = 2 and a set every query outgrows, cmps and results stay the oracle's and no row names a node twice."""
    from roargraph_amd import io
    base, q, off, nbrs, ep = small_set("ip", 4000, 200)
    lists = [nbrs[int(off[i]):int(off[i + 1])].copy() for i in range(base.shape[0])]
    first = int(lists[ep][0])
    lists[first] = np.concatenate([lists[first][:6], lists[first][:6], lists[first][6:]]).astype(np.uint32)
    for i in range(0, 4000, 5):
        lists[i] = np.concatenate([lists[i], lists[i][:3]]).astype(np.uint32)
    o2, n2 = io.lists_to_csr(lists)
    ix = rg.IndexBipartite.from_arrays(base, o2, n2, ep, metric="ip")
    ix.set("lset_tags", 2)
    ix.set("lset_bytes", lset_bytes)
    for L, k in ((20, 10), (150, 100), (700, 10)):
        want = oracle.search(base, "ip", o2, n2, ep, q, k, L, nthreads=4)
        for rep in range(2):
            got = ix.SearchRoarGraph(q, k, L)
            assert (got[2] == want[2]).all(), ("cmps", L, rep)
            assert (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), L
            assert all(len(set(r.tolist())) == k for r in got[0])
    ix.close()


def test_large_base_goes_up_through_pinned_chunks(rg, oracle):
    """rg_index_open_mem with a 230 MB base, a 20 MB graph (round 5: host -> device through two pinned 64-MiB chunks, rg_mem.hip
    upload_staged -- several chunks, a ragged last one): the searches over what arrived equal the oracle's over the host arrays."""
    rng = np.random.default_rng(11)
    nb, d, deg = 287_001, 200, 8
    base = rng.standard_normal((nb, d)).astype(np.float32)
    nbrs = rng.integers(0, nb, size=nb * deg, dtype=np.uint32)
    nbrs.reshape(nb, deg)[:, 0] = (np.arange(nb, dtype=np.uint32) + 1) % nb       # a ring: every node reachable
    off = (np.arange(nb + 1, dtype=np.uint64) * deg)
    q = (rng.standard_normal((24, d)) * 0.5 + 0.3).astype(np.float32)
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, 7, metric="ip")
    for k, L in ((10, 30), (10, 300)):
        got = ix.SearchRoarGraph(q, k, L)
        want = oracle.search(base, "ip", off, nbrs, 7, q, k, L, nthreads=8)
        assert (got[2] == want[2]).all() and (got[3] == want[3]).all() and (got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all(), (k, L)
    ids = np.arange(0, nb, 1009, dtype=np.uint32)      # rows from every chunk, the last one included
    assert (bits(ix.score_batch(q[0], ids)) == bits(oracle.score_batch(base, "ip", q[0], ids))).all()
    ix.close()
