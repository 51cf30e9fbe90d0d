"""TEST INFRASTRUCTURE ONLY (oracle/): a CPU ground-truth builder in the shape of the reference's
`compute_groundtruth` (DiskANN utility, README.md:62-75: blocked fp32 SGEMM on all host cores + per-query top-K
selection; the DiskANN sources themselves are an empty submodule in /root/reference, so this restates the published
algorithm: inner products by GEMM, L2 through |q|^2 + |b|^2 - 2 q.b, K best kept per query across base blocks).

Used by bench.py's cpu_baseline of the ground-truth leg and checked against the fp64 oracle in tests/.  Never imported
by the product."""
import numpy as np


def groundtruth_blocked(base, queries, metric, K, block=131072):
    """Returns (ids uint32 [nq,K], dists float32 [nq,K]) ordered best first: IP -> largest dot first (stored as +dot,
    test_search_bipartite.cpp:46-48 convention), L2 -> smallest squared distance first."""
    base = np.ascontiguousarray(base, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    nq, nb = queries.shape[0], base.shape[0]
    l2 = metric == "l2"
    qn = (queries * queries).sum(1) if l2 else None
    best_s = None   # ranking value, larger = better
    best_i = None
    for s in range(0, nb, block):
        b = base[s:s + block]
        sc = queries @ b.T                                   # SGEMM
        if l2:
            sc = 2.0 * sc - (b * b).sum(1)[None, :]          # larger = closer (|q|^2 is constant per query)
        k = min(K, sc.shape[1])
        part = np.argpartition(-sc, k - 1, axis=1)[:, :k]
        ps = np.take_along_axis(sc, part, axis=1)
        pi = (part + s).astype(np.int64)
        if best_s is None:
            best_s, best_i = ps, pi
        else:
            cs = np.concatenate([best_s, ps], axis=1)
            ci = np.concatenate([best_i, pi], axis=1)
            kk = min(K, cs.shape[1])
            sel = np.argpartition(-cs, kk - 1, axis=1)[:, :kk]
            best_s = np.take_along_axis(cs, sel, axis=1)
            best_i = np.take_along_axis(ci, sel, axis=1)
    order = np.argsort(-best_s, axis=1, kind="stable")
    best_s = np.take_along_axis(best_s, order, axis=1)
    best_i = np.take_along_axis(best_i, order, axis=1)
    dists = (qn[:, None] - best_s) if l2 else best_s
    return best_i.astype(np.uint32), dists.astype(np.float32)
