"""-m gpu: the BASELINE.json configurations at their own shapes.

  config 5  webvid-shaped d = 512 inner product, end to end: ground truth (K2) -> graph construction -> search (K1),
            every leg checked (fp64 brute force for the truth, the CPU oracle's SearchRoarGraph over the SAME index
            for the search), plus the query-sharded search over index replicas
            (tests/test_build_roargraph.cpp:117-136 and tests/test_search_roargraph.cpp:160-209 are the call
            sequences this mirrors)
  config 2  t2i-10M-shaped: 10M x 200 inner product, top-10, L_pq = 500 -- a 256-query sample of the HIP path against
            the oracle on the full-size base (a parity failure at 10M is a red test here, not a string in bench.py)
  config 4  LAION-shaped ground truth: unit-norm clustered embeddings, L2, K = 100, at 1M rows, where the rank-100 /
            rank-101 gap is of the order of the fp32 rounding of the ranking value
  config 3  compute_groundtruth at its own size: 10M x 200 MIPS, K = 100 -- 10,000 queries over the whole base (the balanced work
            split, thresholds and compactions of K2 at the size the bench runs), over one 1.25M-row shard with a non-zero id base
            (what one of eight GPUs holds) and over eight shards + K3, a 256-query sample of each against fp64 brute force
"""
import os

import numpy as np
import pytest

from helpers import bits
from test_gpu_groundtruth import check_gt

pytestmark = pytest.mark.gpu


def _search_equal(got, want, cmps=True):
    assert (got[3] == want[3]).all(), "hops differ"
    assert (got[0] == want[0]).all(), "neighbour ids differ"
    assert (bits(got[1]) == bits(want[1])).all(), "distance bits differ"
    if cmps:
        assert (got[2] == want[2]).all(), "cmps differ"


@pytest.mark.parametrize("gpu_build", [False, True])
def test_config5_d512_ip_end_to_end(oracle, gpu_build):
    from roargraph_amd import build, groundtruth, index, synth
    metric, d, nb, ntrain, nq = "ip", 512, 6000, 1500, 96
    base, train = synth.make_synth(2025, nb, ntrain, d)
    q = synth.make_synth(2026, nb, nq, d)[1]
    # 1. ground truth of the training queries (the bipartite graph's input), K = 100 as in README.md:70-74
    tr_ids, tr_d = groundtruth.compute_groundtruth(base, train, metric, 100)
    ref_ids, _, ref_s = oracle.groundtruth_f64(base, train, metric, 100, nthreads=16)
    check_gt(base, train, metric, 100, tr_ids, tr_d, ref_ids, ref_s)
    # 2. graph construction at d = 512 (paper parameters, README.md:92-97)
    off, nbrs, ep = build.build_roargraph(base, tr_ids, metric, 100, 35, 500, num_threads=1 if not gpu_build else 8,
                                          device=0 if gpu_build else None)
    deg = np.diff(off.astype(np.int64))
    assert off.shape[0] == nb + 1 and int(off[-1]) == nbrs.shape[0] and ep < nb
    assert deg.max() <= 70 and deg.mean() > 10 and (nbrs < nb).all()          # <= 2 * M_pjbp, no dangling ids
    for i in (0, 17, nb - 1):
        row = nbrs[int(off[i]):int(off[i + 1])]
        assert len(set(row.tolist())) == row.shape[0] and i not in row          # no duplicate edges, no self loops
    # 3. search: HIP against the oracle over the SAME index, several beam widths, top-10 and top-100
    ix = index.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    ev_ids, ev_d = groundtruth.compute_groundtruth(base, q, metric, 100)
    ref_ids, _, ref_s = oracle.groundtruth_f64(base, q, metric, 100, nthreads=16)
    check_gt(base, q, metric, 100, ev_ids, ev_d, ref_ids, ref_s)
    recalls = {}
    for k, L in ((10, 20), (10, 100), (10, 500), (100, 200)):
        got = ix.SearchRoarGraph(q, k, L)
        want = oracle.search(base, metric, off, nbrs, ep, q, k, L, nthreads=8)
        _search_equal(got, want)
        recalls[(k, L)] = index.recall(got[0], ev_ids, 10)
        assert recalls[(k, L)] == pytest.approx(oracle.recall(want[0][:, :10].copy(), ref_ids, 10), abs=1e-6)
    assert recalls[(10, 500)] > 0.9 and recalls[(10, 500)] >= recalls[(10, 20)]
    # 4. config 5's search leg: index replicated, queries sharded (two replicas on the one GPU of the test box)
    rep = index.IndexBipartite.from_arrays(base, off, nbrs, ep, metric=metric)
    got = index.search_sharded([ix, rep], q, 10, 100)
    _search_equal(got, oracle.search(base, metric, off, nbrs, ep, q, 10, 100, nthreads=8))
    rep.close()
    ix.close()


def test_config2_10m_x_200_parity_sample(oracle):
    """10M x 200 IP, top-10, L_pq = 500: 256 queries of a 4,096-query batch, every output against the oracle run on the
    same full-size inputs (and, round 3, L_pq = 2000: 48 queries of a 1,024-query batch in every form of the visited set).  The adjacency mixes uniformly random edges with edges to nearby ids, so that nodes are met
    again and again (the LDS filter forgets, the id log + exact distinct count has work to do) -- a random graph alone
    never revisits anything at this size."""
    import torch
    from roargraph_amd.index import IndexBipartite
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(20252)
    nb, d, deg, nq, k, L, ns = 10_000_000, 200, 40, 4096, 10, 500, 256
    base = torch.empty((nb, d), device=dev)
    for s in range(0, nb, 1 << 20):
        base[s:s + (1 << 20)].normal_(generator=g)
    nbrs = torch.randint(0, nb, (nb, deg), dtype=torch.int64, device=dev, generator=g)
    near = (torch.arange(nb, device=dev)[:, None] + torch.randint(-48, 49, (nb, deg // 2), device=dev, generator=g)).clamp_(0, nb - 1)
    nbrs[:, : deg // 2] = near
    nbrs = nbrs.to(torch.int32).reshape(-1)
    off = torch.arange(0, nb + 1, dtype=torch.int64, device=dev) * deg
    q = torch.empty((nq, d), device=dev).normal_(generator=g) * 0.5 + 0.3
    ix = IndexBipartite.from_device(base, off, nbrs, 12345, metric="ip")
    outs = {}
    for vis in (2, 0):
        ix.set("visited", vis)
        ids = torch.zeros((nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((nq, k), device=dev)
        cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
        ix.search_dev(q, k, L, ids, ds, cm, hp); ix.search_wait()
        outs[vis] = tuple(x.cpu().numpy() for x in (ids, ds, cm, hp))
    # round 3: a wide beam at full size -- the look-ahead form of the exact words (automatic from L_pq = 1200), the returning
    # atomics, and the filter + log form, 1,024 queries each at L_pq = 2000
    L2, nq2, ns2 = 2000, 1024, 48
    wide = {}
    for name, knobs in (("look", {"visited": 0, "lookahead": 1}), ("atomics", {"visited": 0, "lookahead": 0}), ("log", {"visited": 2, "lookahead": -1})):
        for kn, v in knobs.items():
            ix.set(kn, v)
        ids = torch.zeros((nq2, k), dtype=torch.int32, device=dev); ds = torch.zeros((nq2, k), device=dev)
        cm = torch.zeros(nq2, dtype=torch.int32, device=dev); hp = torch.zeros(nq2, dtype=torch.int32, device=dev)
        ix.search_dev(q[:nq2], k, L2, ids, ds, cm, hp); ix.search_wait()
        wide[name] = tuple(x.cpu().numpy() for x in (ids, ds, cm, hp))
    ix.close()
    hb, hq = base.cpu().numpy(), q[:ns].cpu().numpy()
    hoff, hn = off.cpu().numpy().view(np.uint64), nbrs.cpu().numpy().view(np.uint32)
    del base, nbrs, near
    want = oracle.search(hb, "ip", hoff, hn, 12345, hq, k, L, nthreads=min(32, os.cpu_count() or 1))
    assert want[2].mean() > 2000, "the sample should be a long search"
    for vis in (2, 0):
        got = tuple(x[:ns] for x in outs[vis])
        _search_equal((got[0].view(np.uint32), got[1], got[2].view(np.uint32), got[3].view(np.uint32)), want)
    # the whole batch: both exact visited forms agree bit for bit
    for a, b in zip(outs[2], outs[0]):
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
    want2 = oracle.search(hb, "ip", hoff, hn, 12345, hq[:ns2], k, L2, nthreads=min(32, os.cpu_count() or 1))
    for name in ("look", "atomics", "log"):
        got = tuple(x[:ns2] for x in wide[name])
        _search_equal((got[0].view(np.uint32), got[1], got[2].view(np.uint32), got[3].view(np.uint32)), want2)
        for a, b in zip(wide[name], wide["look"]):
            assert (a.view(np.uint32) == b.view(np.uint32)).all(), name


def test_config4_gt_unit_norm_clustered_l2_k100(oracle):
    """Unit-norm clustered embeddings (CLIP-like), L2, K = 100, 1M rows: neighbours of a query sit in one tight cluster,
    so the gap between rank 100 and rank 101 is of the order of the fp32 rounding of q.b - |b|^2/2.  K2 keeps a margin
    of survivors past K through the exact re-score (rg_gt_rescore_kernel); every returned id must be a member of the
    fp64 top-100 up to the tie band."""
    import torch
    from roargraph_amd import groundtruth
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(404)
    nb, d, ncl, nq, K = 1_000_000, 512, 1000, 128, 100
    centers = torch.nn.functional.normalize(torch.empty((ncl, d), device=dev).normal_(generator=g), dim=1)
    assign = torch.randint(0, ncl, (nb,), device=dev, generator=g)
    base = centers[assign] + 0.05 * torch.empty((nb, d), device=dev).normal_(generator=g) / d ** 0.5
    base = torch.nn.functional.normalize(base, dim=1).contiguous()
    qa = torch.randint(0, ncl, (nq,), device=dev, generator=g)
    q = torch.nn.functional.normalize(centers[qa] + 0.05 * torch.empty((nq, d), device=dev).normal_(generator=g) / d ** 0.5, dim=1).contiguous()
    ids = torch.zeros((nq, K), dtype=torch.int32, device=dev); vals = torch.zeros((nq, K), device=dev)
    groundtruth.gt_shard_dev(base, q, "l2", K, 0, ids, vals); torch.cuda.synchronize()
    hb, hq = base.cpu().numpy(), q.cpu().numpy()
    ref_ids, _, ref_s = oracle.groundtruth_f64(hb, hq, "l2", K, nthreads=min(32, os.cpu_count() or 1))
    gap = ref_s[:, -1] - ref_s[:, -2]
    assert np.median(gap) < 1e-3 * ref_s[:, -1].mean(), "the set is meant to have tight rank-K boundaries"
    check_gt(hb, hq, "l2", K, ids.cpu().numpy().view(np.uint32), vals.cpu().numpy(), ref_ids, ref_s)
    # and sharded over three row ranges + K3: the same lists
    parts_i = torch.zeros((3, nq, K), dtype=torch.int32, device=dev); parts_v = torch.zeros((3, nq, K), device=dev)
    for r, (lo, hi) in enumerate(groundtruth.shard_rows(nb, 3)):
        groundtruth.gt_shard_dev(base[lo:hi], q, "l2", K, lo, parts_i[r], parts_v[r])
    mi = torch.zeros_like(ids); mv = torch.zeros_like(vals)
    groundtruth.gt_merge_dev(parts_i, parts_v, 3, nq, K, "l2", mi, mv); torch.cuda.synchronize()
    assert torch.equal(mi, ids) and torch.equal(mv.view(torch.int32), vals.view(torch.int32))


def _full_size_sample(oracle, ix, base, off, nbrs, ep, q, metric, cases, ns, knob_sets):
    """every (k, L) of `cases` through the device form on the whole batch in every knob set, then the first `ns` queries
    against the oracle over the same full-size inputs; all knob sets must agree on the whole batch bit for bit"""
    import torch
    dev = q.device
    nq = q.shape[0]
    outs = {}
    for name, knobs in knob_sets:
        for kn, v in knobs.items():
            ix.set(kn, v)
        for k, L in cases:
            ids = torch.zeros((nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((nq, k), device=dev)
            cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
            ix.search_dev(q, k, L, ids, ds, cm, hp); ix.search_wait()
            outs[(name, k, L)] = tuple(x.cpu().numpy() for x in (ids, ds, cm, hp))
    hb, hq = base.cpu().numpy(), q[:ns].cpu().numpy()
    hoff, hn = off.cpu().numpy().view(np.uint64), nbrs.cpu().numpy().view(np.uint32)
    for k, L in cases:
        want = oracle.search(hb, metric, hoff, hn, ep, hq, k, L, nthreads=min(32, os.cpu_count() or 1))
        first = knob_sets[0][0]
        for name, _ in knob_sets:
            got = tuple(x[:ns] for x in outs[(name, k, L)])
            _search_equal((got[0].view(np.uint32), got[1], got[2].view(np.uint32), got[3].view(np.uint32)), want)
            for a, b in zip(outs[(name, k, L)], outs[(first, k, L)]):
                assert (a.view(np.uint32) == b.view(np.uint32)).all(), (name, k, L)
    return outs


def test_config4_10m_x_512_l2_top100_parity_sample(oracle):
    """BASELINE configs[3] at full size: 10M x 512, squared L2, top-100 -- 96 queries of a 2,048-query batch at L_pq = 200 and
    1000 against the oracle over the same 20.5 GB base, in the default visited mode, on the exact tags and on the filter
    alone (ids / distances / hops; the whole batch equal between the forms).  The adjacency mixes uniformly random edges
    with edges to nearby ids (nodes are met again and again), as in the d = 200 test above."""
    import torch
    from roargraph_amd.index import IndexBipartite
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(20254)
    nb, d, deg, nq, ns = 10_000_000, 512, 40, 2048, 96
    base = torch.empty((nb, d), device=dev)
    for s in range(0, nb, 1 << 19):
        base[s:s + (1 << 19)].normal_(generator=g)
    nbrs = torch.randint(0, nb, (nb, deg), dtype=torch.int64, device=dev, generator=g)
    near = (torch.arange(nb, device=dev)[:, None] + torch.randint(-48, 49, (nb, deg // 2), device=dev, generator=g)).clamp_(0, nb - 1)
    nbrs[:, : deg // 2] = near
    del near
    nbrs = nbrs.to(torch.int32).reshape(-1)
    off = torch.arange(0, nb + 1, dtype=torch.int64, device=dev) * deg
    q = torch.empty((nq, d), device=dev).normal_(generator=g) * 0.5 + 0.3
    ix = IndexBipartite.from_device(base, off, nbrs, 4321, metric="l2")
    _full_size_sample(oracle, ix, base, off, nbrs, 4321, q, "l2", ((100, 200), (100, 1000)), ns,
                      (("default", {"visited": 2}), ("exact_tags", {"visited": 0})))
    ix.close()


def test_config5_2p5m_x_512_ip_product_built_index(oracle):
    """BASELINE configs[4] at full size, over an index the product built: 2.5M x 512 inner product, ground truth of 500k training
    queries (K2) -> GPU-assisted construction (M_sq = 100, M_pjbp = 35, L_pjpq = 500) -> 2,048 queries, top-10, L_pq = 50 and
    500; the first 128 against the oracle over the same index, default mode and exact tags; the eval-side ground truth of those
    128 against fp64."""
    import torch
    from roargraph_amd import build, groundtruth, synth
    from roargraph_amd.index import IndexBipartite
    dev = torch.device("cuda", 0)
    nb, d, ntrain, nq, ns = 2_500_000, 512, 500_000, 2048, 128
    base, train, q, _ = synth.make_device_set(dev, 77, nb, ntrain, nq, d, data="lowrank", rank=32, q_seed=5)
    ti, _ = groundtruth.groundtruth_distributed(base, 0, train, "ip", 100)
    torch.cuda.synchronize()
    h_off, h_nbrs, ep = build.build_roargraph(base.cpu().numpy(), ti.cpu().numpy().view(np.uint32), "ip", 100, 35, 500,
                                              num_threads=min(64, os.cpu_count() or 1), device=0)
    del train, ti
    deg = np.diff(h_off.astype(np.int64))
    assert deg.max() <= 70 and deg.mean() > 8 and int(h_nbrs.max()) < nb
    off = torch.from_numpy(h_off.view(np.int64)).to(dev)
    nbrs = torch.from_numpy(h_nbrs.view(np.int32)).to(dev)
    ix = IndexBipartite.from_device(base, off, nbrs, ep, metric="ip")
    outs = _full_size_sample(oracle, ix, base, off, nbrs, ep, q, "ip", ((10, 50), (10, 500)), ns,
                             (("default", {"visited": 2}), ("exact_tags", {"visited": 0})))
    ix.close()
    # recall of the sample against fp64 truth: a genuine index over structured data reaches 0.9 well before L_pq = 500
    gi = torch.zeros((ns, 100), dtype=torch.int32, device=dev); gv = torch.zeros((ns, 100), device=dev)
    groundtruth.gt_shard_dev(base, q[:ns].contiguous(), "ip", 100, 0, gi, gv); torch.cuda.synchronize()
    hb, hq = base.cpu().numpy(), q[:ns].cpu().numpy()
    ref_ids, _, ref_s = oracle.groundtruth_f64(hb, hq, "ip", 100, nthreads=min(32, os.cpu_count() or 1))
    check_gt(hb, hq, "ip", 100, gi.cpu().numpy().view(np.uint32), gv.cpu().numpy(), ref_ids, ref_s)
    from roargraph_amd import index as ixmod
    rec = ixmod.recall(outs[("default", 10, 500)][0][:ns].view(np.uint32), ref_ids, 10)
    assert rec > 0.95, rec


def test_config3_gt_10m_x_200_mips_k100_full_size(oracle):
    """BASELINE configs[2] at full size on one GPU: K2 over 10M x 200 rows, MIPS, K = 100, 10,000 queries (474 work
    items on 512 workgroups: the balanced split, threshold filter, candidate buffers and compactions all busy), the first 256
    queries against the fp64 brute force of the checker over the same 8 GB base; one 1.25M-row shard with id_base = 3 x 1.25M (the
    fourth of eight GPUs) against fp64 over that slice; eight shards + K3 == the one-shot lists bit for bit
    (/root/reference/src/index_bipartite.cpp:2622-2642 is the consumer of these lists)."""
    import torch
    from roargraph_amd import groundtruth, synth
    dev = torch.device("cuda", 0)
    nb, d, nq, ns, K = 10_000_000, 200, 10_000, 256, 100
    base, _, q, _ = synth.make_device_set(dev, 1234, nb, 0, nq, d, data="lowrank", rank=32, q_seed=4711)
    ids = torch.zeros((nq, K), dtype=torch.int32, device=dev); vals = torch.zeros((nq, K), device=dev)
    groundtruth.gt_shard_dev(base, q, "ip", K, 0, ids, vals); torch.cuda.synchronize()
    hb, hq = base.cpu().numpy(), q[:ns].cpu().numpy()
    nt = min(64, os.cpu_count() or 1)
    ref_ids, _, ref_s = oracle.groundtruth_f64(hb, hq, "ip", K, nthreads=nt)
    check_gt(hb, hq, "ip", K, ids[:ns].cpu().numpy().view(np.uint32), vals[:ns].cpu().numpy(), ref_ids, ref_s)
    # one shard of eight, ids offset by its first row
    shards = groundtruth.shard_rows(nb, 8)
    lo, hi = shards[3]
    assert lo == 3 * 1_250_000 and hi - lo == 1_250_000
    si = torch.zeros((nq, K), dtype=torch.int32, device=dev); sv = torch.zeros((nq, K), device=dev)
    groundtruth.gt_shard_dev(base[lo:hi], q, "ip", K, lo, si, sv); torch.cuda.synchronize()
    r_ids, _, r_s = oracle.groundtruth_f64(hb[lo:hi], hq, "ip", K, nthreads=nt)
    got = si[:ns].cpu().numpy().view(np.uint32)
    assert got.min() >= lo and got.max() < hi
    check_gt(hb[lo:hi], hq, "ip", K, got - np.uint32(lo), sv[:ns].cpu().numpy(), r_ids, r_s)
    # eight shards + K3: the one-shot lists, bit for bit (ids and ranking values)
    parts_i = torch.zeros((8, nq, K), dtype=torch.int32, device=dev); parts_v = torch.zeros((8, nq, K), device=dev)
    for r, (a, b) in enumerate(shards):
        groundtruth.gt_shard_dev(base[a:b], q, "ip", K, a, parts_i[r], parts_v[r])
    assert torch.equal(parts_i[3], si) and torch.equal(parts_v[3].view(torch.int32), sv.view(torch.int32))
    mi = torch.zeros_like(ids); mv = torch.zeros_like(vals)
    groundtruth.gt_merge_dev(parts_i, parts_v, 8, nq, K, "ip", mi, mv); torch.cuda.synchronize()
    same = mi == ids
    if not bool(same.all()):      # rows may differ only inside exact fp32 ties of the ranking value (equal bits, ids swapped)
        assert torch.equal(mv.view(torch.int32), vals.view(torch.int32)), "K3 over eight shards returned other values than one pass"
        assert float(same.float().mean()) > 0.9999
    else:
        assert torch.equal(mv.view(torch.int32), vals.view(torch.int32))
