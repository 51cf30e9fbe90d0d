"""Seeded synthetic inputs for tests and the bench.

No dataset can be downloaded here, so every input is generated: base vectors
~ N(0,1)^d, queries ~ N(0.3, 0.5^2)^d (an out-of-distribution shift, the
cross-modal situation RoarGraph targets).  Graphs come from two generators:

  knn_graph      exact metric k-NN lists + reverse edges + a few random
                 long-range edges (small sets: fixtures, parity tests)
  random_regular uniformly random out-neighbours (10M-node bench graph: the
                 same HBM access pattern as a real index; recall is
                 meaningless on it and is reported as such)

Neither is RoarGraph's construction algorithm (src/index_bipartite.cpp:1043-1277,
a CPU path that SURVEY.md section 8(f) lists as "next"); a .index file is just an
INPUT of the search path, so any adjacency structure exercises it.
"""
import numpy as np


def make_synth(seed, nb, nq, d, dtype=np.float32):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((nb, d), dtype=np.float32)
    queries = (0.3 + 0.5 * rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    return base.astype(dtype), queries.astype(dtype)


def _scores(a, b, metric):
    """smaller = closer, float64"""
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    if metric == "l2":
        return (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T
    return -(a @ b.T)


def centroid_entry_point(base):
    """argmin squared L2 to the mean vector, the rule of CalculateProjectionep (src/index_bipartite.cpp:2004-2041)."""
    c = base.astype(np.float64).mean(0)
    return int(((base.astype(np.float64) - c) ** 2).sum(1).argmin())


def knn_graph(base, metric, M=16, n_random=4, max_deg=None, seed=7, train_queries=None, block=2048):
    """Adjacency lists (python lists of uint32 arrays) + entry point.

    If train_queries is given, neighbours are linked the way a bipartite projection does in spirit: every
    query's nearest base point is connected to that query's other near neighbours; otherwise base-base k-NN.
    """
    rng = np.random.default_rng(seed)
    nb = base.shape[0]
    adj = [set() for _ in range(nb)]
    if train_queries is not None:
        for s in range(0, train_queries.shape[0], block):
            sc = _scores(train_queries[s:s + block], base, metric)
            top = np.argsort(sc, axis=1, kind="stable")[:, :M]
            for row in top:
                p = int(row[0])
                for x in row[1:]:
                    adj[p].add(int(x))
                    adj[int(x)].add(p)
    for s in range(0, nb, block):
        sc = _scores(base[s:s + block], base, metric)
        sc[np.arange(sc.shape[0]), np.arange(s, s + sc.shape[0])] = np.inf
        top = np.argsort(sc, axis=1, kind="stable")[:, :M]
        for i, row in enumerate(top):
            for x in row:
                adj[s + i].add(int(x))
                adj[int(x)].add(s + i)
    for i in range(nb):
        for x in rng.integers(0, nb, n_random):
            if int(x) != i:
                adj[i].add(int(x))
    max_deg = max_deg or 4 * M
    lists = []
    for i in range(nb):
        l = np.array(sorted(adj[i]), np.uint32)
        if l.shape[0] > max_deg:
            l = rng.permutation(l)[:max_deg]
        lists.append(rng.permutation(l).astype(np.uint32))
    return lists, centroid_entry_point(base)


def random_regular_csr(nd, deg, seed=11):
    rng = np.random.default_rng(seed)
    nbrs = rng.integers(0, nd, size=(nd, deg), dtype=np.uint32)
    offsets = (np.arange(nd + 1, dtype=np.uint64) * np.uint64(deg))
    return offsets, nbrs.reshape(-1)
