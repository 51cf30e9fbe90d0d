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


# the "mixture" family: (cluster centres, spread of the base rows around their centre, shift and spread of the queries) in the latent space
# (round 6, scripts/r06/mixture_scan.py at 1M rows: 1,000 tight clusters (0.35) give a degenerate index -- average degree 3.8 at 10M rows, a fifth
# of the nodes with two edges or fewer -- 10,000 wider ones an index of degree 33 like a real one; every variant is EASY for a graph search,
# recall@10 >= 0.99 at L_pq 50: what the family offers is low reuse between the queries of a launch, not difficulty)
MIXTURE_DEFAULT = (10000, 0.5, 0.2, 0.6)


def make_device_set(dev, seed, nb, ntrain, nq, d, data="gaussian", rank=24, noise=0.05, q_seed=None):
    """Base / training queries / test queries as torch tensors on `dev` (bench.py, scripts/e2e_pipeline.py).

    "gaussian": base ~ N(0,1)^d, queries ~ N(0.3, 0.5^2)^d -- the hardest case (no structure: a graph index cannot
                reach high recall on it at 10M points; used for throughput).
    "lowrank":  embeddings with the structure real ones have, a low intrinsic dimension: x = z A + noise * eps with
                z ~ N(0, I_rank) for the base and z ~ N(0.3, 0.5^2 I_rank) for the queries (the same out-of-distribution
                shift, in the latent space), A a fixed rank x d matrix with N(0, 1/rank) entries.  This is the set the
                recall target (>= 0.9 recall@10) is demonstrated on.
    q_seed: the test queries come from their own generator seeded with it (one batch per rank over the same base).
    Returns (base, train, queries, description)."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    chunk = 1 << 20

    def fill(n, mean, std, mix):
        out = torch.empty((n, d), dtype=torch.float32, device=dev)
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            if mix is None:
                out[s:s + m].normal_(generator=g)
                if mean != 0.0 or std != 1.0:
                    out[s:s + m].mul_(std).add_(mean)
            else:
                z = torch.empty((m, rank), dtype=torch.float32, device=dev).normal_(generator=g) * std + mean
                out[s:s + m] = z @ mix
                out[s:s + m].add_(torch.empty((m, d), dtype=torch.float32, device=dev).normal_(generator=g), alpha=noise)
        return out

    def queries(mix):
        nonlocal g
        if q_seed is not None:
            g = torch.Generator(device=dev)
            g.manual_seed(q_seed)
        return fill(nq, 0.3, 0.5, mix)

    if data == "gaussian":
        base = fill(nb, 0.0, 1.0, None)
        train = fill(ntrain, 0.3, 0.5, None) if ntrain else None
        q = queries(None)
        desc = "base N(0,1) %dx%d, train/test queries N(0.3,0.5^2) (%d / %d)" % (nb, d, ntrain, nq)
    elif data == "lowrank":
        mix = torch.empty((rank, d), dtype=torch.float32, device=dev).normal_(generator=g) / float(rank) ** 0.5
        base = fill(nb, 0.0, 1.0, mix)
        train = fill(ntrain, 0.3, 0.5, mix) if ntrain else None
        q = queries(mix)
        desc = ("low-rank embeddings %dx%d: x = zA + %.2f eps, latent rank %d, base z ~ N(0,1), train/test queries "
                "z ~ N(0.3,0.5^2) (%d / %d)" % (nb, d, noise, rank, ntrain, nq))
    elif data == "mixture":
        # (round 6) low reuse BETWEEN queries: cluster centres in the latent space, a row = its centre + spread x N(0, I) mapped through A;
        # queries sit near centres too, with the out-of-distribution shift of the other families.  Two queries of a batch rarely walk the
        # same region of the graph, so few of a launch's row reads are repeats that the Infinity Cache can serve.
        import os
        ncl, b_spread, q_shift, q_spread = MIXTURE_DEFAULT
        if os.environ.get("RG_MIXTURE"):      # experiments: "clusters,base spread,query shift,query spread"
            v = os.environ["RG_MIXTURE"].split(",")
            ncl, b_spread, q_shift, q_spread = int(v[0]), float(v[1]), float(v[2]), float(v[3])
        mix = torch.empty((rank, d), dtype=torch.float32, device=dev).normal_(generator=g) / float(rank) ** 0.5
        cent = torch.empty((ncl, rank), dtype=torch.float32, device=dev).normal_(generator=g)

        def fill_mix(n, shift, spread):
            out = torch.empty((n, d), dtype=torch.float32, device=dev)
            for s in range(0, n, chunk):
                m = min(chunk, n - s)
                k = torch.randint(0, ncl, (m,), device=dev, generator=g)
                z = cent[k] + torch.empty((m, rank), dtype=torch.float32, device=dev).normal_(generator=g) * spread + shift
                out[s:s + m] = z @ mix
                out[s:s + m].add_(torch.empty((m, d), dtype=torch.float32, device=dev).normal_(generator=g), alpha=noise)
            return out
        base = fill_mix(nb, 0.0, b_spread)
        train = fill_mix(ntrain, q_shift, q_spread) if ntrain else None
        if q_seed is not None:
            g = torch.Generator(device=dev)
            g.manual_seed(q_seed)
        q = fill_mix(nq, q_shift, q_spread)
        desc = ("mixture embeddings %dx%d: %d cluster centres in a rank-%d latent space, x = (c_k + %.2f e) A + %.2f eps; train/test queries "
                "(c_k + %.2f + %.2f e) A (%d / %d)" % (nb, d, ncl, rank, b_spread, noise, q_shift, q_spread, ntrain, nq))
    else:
        raise ValueError("data must be gaussian, lowrank or mixture")
    return base, train, q, desc


_PIN = {}


def _pinned(torch, nbytes=64 << 20):
    """One pinned staging buffer per process (bench / tests helper)."""
    if "buf" not in _PIN:
        _PIN["buf"] = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    return _PIN["buf"]


def to_host(t):
    """Device tensor -> numpy array through a pinned 64-MiB staging buffer (round 5).  `t.cpu()` of gigabytes hands pageable memory to
    the runtime, which pins it page by page for the copy; two bench runs of the round died of a GPU memory access fault a few pages
    into such a region.  Small tensors take the plain path."""
    import numpy as np
    import torch
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if not t.is_cuda or nbytes < (256 << 20):
        return t.cpu().numpy()
    out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
    flat_out = out.reshape(-1).view(np.uint8)
    flat_in = t.reshape(-1).view(torch.uint8)
    pin = _pinned(torch)
    step = pin.numel()
    for off in range(0, nbytes, step):
        n = min(step, nbytes - off)
        pin[:n].copy_(flat_in[off:off + n], non_blocking=False)
        flat_out[off:off + n] = pin[:n].numpy()
    return out


def to_device(a, dev):
    """numpy array -> device tensor through the same pinned staging buffer (see to_host)."""
    import numpy as np
    import torch
    a = np.ascontiguousarray(a)
    if a.nbytes < (256 << 20):
        return torch.from_numpy(a).to(dev)
    out = torch.empty(tuple(a.shape), dtype=torch.from_numpy(a[:0].reshape(-1)[:0]).dtype, device=dev)
    flat_out = out.reshape(-1).view(torch.uint8)
    flat_in = a.reshape(-1).view(np.uint8)
    pin = _pinned(torch)
    step = pin.numel()
    for off in range(0, a.nbytes, step):
        n = min(step, a.nbytes - off)
        pin[:n].numpy()[:] = flat_in[off:off + n]
        flat_out[off:off + n].copy_(pin[:n], non_blocking=False)
    return out
