"""-m gpu: randomized differential test of SearchRoarGraph -- HIP path vs the oracle over random graph shapes, dimensions,
beam widths, k, visited modes and launch knobs (every knob that must not change results is drawn at random too).

Each case is small (the oracle answers in milliseconds); the seeds are fixed, so a failure names a reproducible case."""
import numpy as np
import pytest

from helpers import bits

pytestmark = pytest.mark.gpu

DIMS = [200, 200, 200, 512, 512, 8, 24, 64, 96, 104, 136, 256, 264]


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    d = DIMS[seed % len(DIMS)] if seed < 60 else (200, 512, 200)[seed % 3]
    nb = int(rng.integers(300, 4000))
    metric = "ip" if rng.random() < 0.5 else "l2"
    structured = rng.random() < 0.5
    if structured:      # low-rank rows: neighbourhoods overlap, the visited filter and the de-duplication have work to do
        r = int(rng.integers(2, 12))
        A = (rng.standard_normal((r, d)) / np.sqrt(r)).astype(np.float32)
        base = (rng.standard_normal((nb, r)).astype(np.float32) @ A + 0.05 * rng.standard_normal((nb, d)).astype(np.float32))
        nq = int(rng.integers(1, 70))
        q = ((0.3 + 0.5 * rng.standard_normal((nq, r))).astype(np.float32) @ A).astype(np.float32)
    else:
        base = rng.standard_normal((nb, d)).astype(np.float32)
        nq = int(rng.integers(1, 70))
        q = (0.3 + 0.5 * rng.standard_normal((nq, d))).astype(np.float32)
    if rng.random() < 0.3:                      # exact ties: some rows appear several times
        dup = rng.integers(0, nb, nb // 5)
        base[dup] = base[rng.integers(0, nb, nb // 5)]
    maxdeg = int(rng.choice([3, 8, 20, 40, 70, 130]))
    deg = rng.integers(1, maxdeg + 1, nb)
    deg[rng.integers(0, nb, nb // 20)] = 0      # dead ends
    ep = int(rng.integers(0, nb))
    deg[ep] = max(deg[ep], min(maxdeg, 5))
    off = np.zeros(nb + 1, np.uint64); off[1:] = np.cumsum(deg)
    if structured:                              # edges to near rows (by a random projection) + a few random ones
        key = base @ rng.standard_normal(d).astype(np.float32)
        order = np.argsort(key); pos = np.empty(nb, np.int64); pos[order] = np.arange(nb)
        nbrs = np.empty(int(off[-1]), np.uint32)
        for i in range(nb):
            k = int(deg[i])
            near = order[np.clip(pos[i] + rng.integers(-40, 41, k), 0, nb - 1)]
            far = rng.integers(0, nb, k)
            nbrs[int(off[i]):int(off[i]) + k] = np.where(rng.random(k) < 0.8, near, far)
    else:
        nbrs = rng.integers(0, nb, int(off[-1])).astype(np.uint32)   # repeats and self loops included
    L = int(rng.choice([1, 5, 10, 33, 64, 65, 100, 257, 600]))
    k = int(rng.integers(1, min(L, 100) + 1))
    knobs = {"visited": int(rng.integers(0, 3)),
             "rows_per_pass": int(rng.choice([0, 0, 4, 8, 16, 32])),
             "filter_log2": int(rng.choice([0, 0, 4, 9, 12])),
             "waves_per_cu": int(rng.choice([0, 0, 1, 3])),
             "split_rows": int(rng.random() < 0.7),
             "exact_filter": int(rng.random() < 0.7),
             "query_in_lds": int(rng.random() < 0.2),
             "log_cap": int(rng.choice([0, 0, 64, 1024]))}
    knobs["_csr"] = int(rng.random() < 0.2)     # adjacency layout (read from the environment at open)
    knobs["lookahead"] = int(rng.random() < 0.7)
    knobs["gather_form"] = int(rng.random() < 0.5)
    knobs["filter_min_indeg"] = int(rng.choice([0, 0, 2, 4, 12, 255]))
    knobs["count_in_k1"] = int(rng.choice([-1, -1, 0, 2000]))
    knobs["shared_frontier"] = int(rng.random() < 0.3)
    # lists without repeated ids (what every real index has): the look-ahead form of the exact words applies to them.
    # Cases 60+ aim at it: register-staged dimensions, exact words, ELL rows, beams wide enough to run for a while
    if seed >= 60 or rng.random() < 0.5:
        rows = []
        for i in range(nb):
            r = nbrs[int(off[i]):int(off[i + 1])]
            _, first = np.unique(r, return_index=True)
            rows.append(r[np.sort(first)])
        deg = np.array([len(r) for r in rows])
        off = np.zeros(nb + 1, np.uint64); off[1:] = np.cumsum(deg)
        nbrs = np.concatenate(rows).astype(np.uint32) if nb else nbrs
    if seed >= 60:
        knobs.update(visited=0, lookahead=1, _csr=0, query_in_lds=0, rows_per_pass=int(rng.choice([0, 8, 16, 32])))
    knobs["gather_roll"] = int(rng.random() < 0.5)          # (drawn last: the cases of earlier rounds keep their other draws)
    knobs["visited_bytes"] = int(rng.choice([-1, -1, 0]))
    knobs["filter_fill"] = int(rng.choice([1, 2, 2, 0]))
    return base, q, off, nbrs, ep, metric, k, L, knobs


@pytest.mark.parametrize("seed", range(100))
def test_random_case_matches_the_oracle(oracle, seed, monkeypatch):
    from roargraph_amd import index as rg
    base, q, off, nbrs, ep, metric, k, L, knobs = make_case(seed)
    monkeypatch.setenv("RG_FORCE_CSR", str(knobs.pop("_csr")))
    try:
        want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=2)
    except RuntimeError as e:           # "not enough results": the HIP path must refuse the same batch the same way
        ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
        with pytest.raises(Exception, match="not enough results"):
            ix.SearchRoarGraph(q, k, L)
        ix.close()
        assert "not enough results" in str(e)
        return
    ix = rg.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    for name, v in knobs.items():
        ix.set(name, v)
    got = ix.SearchRoarGraph(q, k, L)
    ix.close()
    ctx = (seed, metric, base.shape, k, L, knobs)
    assert (got[0] == want[0]).all(), ("ids", ctx)
    assert (bits(got[1]) == bits(want[1])).all(), ("dists", ctx)
    assert (got[3] == want[3]).all(), ("hops", ctx)
    if knobs["visited"] != 1:
        assert (got[2] == want[2]).all(), ("cmps", ctx)
    else:
        assert (got[2] >= want[2]).all(), ("cmps (evaluations performed)", ctx)
