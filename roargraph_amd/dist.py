"""One-process-per-GPU helpers for the search path (torch.distributed; "nccl" = RCCL on the GPU box, "gloo" in CPU tests).

Search shards by QUERIES: a graph traversal needs the whole graph and every base row, so the index is replicated
(8.0 + 1.9 GB for t2i-10M, far below 288 GB of HBM) and each rank searches a contiguous slice of the query batch --
independent units, no collective on the data path.  The only exchange is the optional all-gather of the k results.
"""
import numpy as np


def query_slice(nq, rank, world):
    per = (nq + world - 1) // world
    return min(nq, rank * per), min(nq, (rank + 1) * per)


def search_sharded(search_fn, queries, k, L_pq, gather=True, group=None):
    """search_fn(queries_slice, k, L_pq) -> (ids, dists, cmps, hops) as numpy arrays (IndexBipartite.SearchRoarGraph).
    Returns this rank's slice, or with gather=True the full batch on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    nq = queries.shape[0]
    lo, hi = query_slice(nq, rank, world)
    ids, dists, cmps, hops = search_fn(queries[lo:hi], k, L_pq)
    if not gather or world == 1:
        return ids, dists, cmps, hops
    per = (nq + world - 1) // world

    def pad(a):
        out = np.zeros((per,) + a.shape[1:], a.dtype)
        out[: a.shape[0]] = a
        return torch.from_numpy(out)

    outs = []
    for a in (ids.astype(np.int64), dists, cmps.astype(np.int64), hops.astype(np.int64)):
        t = pad(a)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        rows = []
        for r in range(world):
            a0, a1 = query_slice(nq, r, world)
            rows.append(parts[r][: a1 - a0])
        outs.append(torch.cat(rows).cpu().numpy())
    return outs[0].astype(np.uint32), outs[1], outs[2].astype(np.uint32), outs[3].astype(np.uint32)


def bench_plan(world, nb, nq, ntrain, gt_nq, gt_batch=65536, K=100, avg_degree=15.0, dim=200):
    """What every rank of `bench.py --gpus world` does, as plain numbers (no GPU needed): the plan the multi-GPU run follows,
    checked on the CPU for world = 8 at the 10M shape (tests/test_dist_gloo.py) and printed by bench.py in its config block.

      search        rank r searches its own batches of nq queries (seeds 99 + r + 7919 b): weak scaling, no collective
      train truth   base rows sharded (groundtruth.shard_rows), every rank scores all ntrain queries against its shard, one
                    all-to-all of the per-shard K-lists, rank r merges the query range it owns; the lists are gathered on all
      index         built on rank 0, broadcast: (nb + 1) offsets of 8 bytes + edges of 4 bytes
      gt_build      rg_groundtruth_rank: per batch of gt_batch queries rank r owns the r-th balanced slice of the batch"""
    from . import groundtruth
    rows = groundtruth.shard_rows(nb, world)
    qr = groundtruth.query_ranges(ntrain, world)
    edges = int(avg_degree * nb)
    plan = {"world": world,
            "search": [{"rank": r, "queries_per_step": nq, "first_batch_seed": 99 + r} for r in range(world)],
            "train_truth": [{"rank": r, "base_rows": list(rows[r]), "owned_queries": list(qr[r]),
                             "all_to_all_send_bytes": int(ntrain) * K * 8 * (world - 1) // world,
                             "scores": (rows[r][1] - rows[r][0]) * int(ntrain)} for r in range(world)],
            "index_broadcast_bytes": (nb + 1) * 8 + edges * 4,
            "replica_bytes_per_gpu": nb * dim * 4 + (nb + 1) * 8 + edges * 4,
            "gt_build": [{"rank": r, "base_rows": list(rows[r]), "owned_rows": int(len(groundtruth.owned_rows(gt_nq, world, r, gt_batch))),
                          "batches": (gt_nq + gt_batch - 1) // gt_batch} for r in range(world)]}
    return plan
