"""Shared builders of small seeded inputs for the parity tests."""
import numpy as np

from roargraph_amd import io, synth

_cache = {}


def small_set(metric, nb, d, nq=64, M=12, seed=1234, ntrain=600):
    key = (metric, nb, d, nq, M, seed, ntrain)
    if key not in _cache:
        base, q = synth.make_synth(seed, nb, nq, d)
        tq = synth.make_synth(seed + 1, nb, ntrain, d)[1] if ntrain else None
        lists, ep = synth.knn_graph(base, metric, M=M, train_queries=tq)
        off, nbrs = io.lists_to_csr(lists)
        _cache[key] = (base, q, off, nbrs, ep)
    return _cache[key]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)
