"""Brute-force k-NN ground truth (the `compute_groundtruth` step of the RoarGraph pipeline, README.md:62-75).

Thin host layer over the C ABI (rg_gt_shard_dev / rg_gt_merge_dev / rg_groundtruth_mem).  The multi-GPU form is one
process per GPU: every rank holds a contiguous ROW SHARD of the base, scores ALL queries against it (K2), then the
per-shard top-K lists are exchanged with one all-to-all (each rank receives the lists of the query range it owns)
and merged (K3).  The exchange goes through torch.distributed -- backend "nccl" is RCCL over xGMI on the GPU box,
"gloo" in the CPU tests.
"""
import ctypes as C

import numpy as np

from ._lib import METRIC, check, lib


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def compute_groundtruth(base, queries, metric, K, devices=None):
    """Host buffers in, (ids[nq,K] u32, dists[nq,K] f32) out; dists = +inner product (mips) or squared L2."""
    base = np.ascontiguousarray(base, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    nq = queries.shape[0]
    ids = np.zeros((nq, K), np.uint32)
    dists = np.zeros((nq, K), np.float32)
    devs = (C.c_int * len(devices))(*devices) if devices else None
    check(lib().rg_groundtruth_mem(_vp(base), C.c_uint32(base.shape[0]), C.c_uint32(base.shape[1]), _vp(queries),
                                   C.c_uint32(nq), C.c_uint32(queries.shape[1]), C.c_uint32(base.shape[1]),
                                   METRIC[metric], C.c_uint32(K), _vp(ids), _vp(dists), devs,
                                   len(devices) if devices else 0))
    return ids, dists


def compute_groundtruth_files(base_fbin, query_fbin, gt_out, metric, K, devices=None):
    devs = (C.c_int * len(devices))(*devices) if devices else None
    check(lib().rg_groundtruth(base_fbin.encode(), query_fbin.encode(), gt_out.encode(), METRIC[metric], C.c_uint32(K),
                               devs, len(devices) if devices else 0))


def gt_shard_dev(base_t, queries_t, metric, K, id_base, ids_t, dists_t, dim=None, stream=0):
    """K2 on torch CUDA tensors (row-major fp32, strides % 4 == 0); async on `stream`."""
    check(lib().rg_gt_shard_dev(C.c_void_p(base_t.data_ptr()), C.c_uint32(base_t.shape[0]), C.c_uint32(base_t.stride(0)),
                                C.c_void_p(queries_t.data_ptr()), C.c_uint32(queries_t.shape[0]),
                                C.c_uint32(queries_t.stride(0)), C.c_uint32(dim or base_t.shape[1]), METRIC[metric],
                                C.c_uint32(K), C.c_uint32(id_base), C.c_void_p(ids_t.data_ptr()),
                                C.c_void_p(dists_t.data_ptr()), base_t.device.index or 0, C.c_void_p(stream)))


def gt_merge_dev(ids_in_t, dists_in_t, nlists, nq, K, metric, ids_t, dists_t, stream=0):
    """K3: [nlists][nq][K] sorted lists -> [nq][K]."""
    check(lib().rg_gt_merge_dev(C.c_void_p(ids_in_t.data_ptr()), C.c_void_p(dists_in_t.data_ptr()), C.c_uint32(nlists),
                                C.c_uint32(nq), C.c_uint32(K), METRIC[metric], C.c_void_p(ids_t.data_ptr()),
                                C.c_void_p(dists_t.data_ptr()), ids_in_t.device.index or 0, C.c_void_p(stream)))


class Comm:
    """rg_comm handles (include/rg.h): the transport of the native multi-rank ground truth.

    Comm.local(devices)            all ranks in this process, one per entry (RCCL when the devices are distinct, peer copies
                                   when ranks share a device); returns a list of Comm
    Comm.from_torch_dist(device)   one process per GPU under torch.distributed: rank 0 makes the RCCL unique id, the
                                   process group broadcasts it, every rank joins (ncclCommInitRank)"""

    def __init__(self, handle, rank, world, device):
        self.handle, self.rank, self.world, self.device = handle, rank, world, device

    @classmethod
    def local(cls, devices):
        n = len(devices)
        arr = (C.c_void_p * n)()
        check(lib().rg_comm_init_local((C.c_int * n)(*devices), n, arr))
        return [cls(C.c_void_p(arr[i]), i, n, devices[i]) for i in range(n)]

    @classmethod
    def from_torch_dist(cls, device, group=None):
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ident = [None]
        if rank == 0 and world > 1:
            buf = C.create_string_buffer(128)
            if lib().rg_comm_unique_id(buf) == 0:      # (a failure travels as None: no rank is left waiting in the broadcast)
                ident = [buf.raw]
        if world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
            if ident[0] is None:
                raise RuntimeError("RCCL is not available on rank 0: no communicator (rg_comm_unique_id failed)")
        h = C.c_void_p()
        check(lib().rg_comm_init_rank(ident[0], rank, world, device, C.byref(h)))
        return cls(h, rank, world, device)

    def uses_rccl(self):
        return bool(lib().rg_comm_uses_rccl(self.handle))

    def destroy(self):
        if self.handle:
            lib().rg_comm_destroy(self.handle)
            self.handle = None


def groundtruth_rank(comm, base_shard_t, id_base, queries, metric, K, out_ids, out_dists, batch=0):
    """rg_groundtruth_rank: this rank's part of the streamed multi-rank ground truth.  base_shard_t: torch CUDA tensor
    (the rank's rows, resident in HBM); queries: host numpy [nq, dim] (same on every rank); out_ids / out_dists: host
    numpy [nq, K] -- the rows this rank owns are written, the others left alone."""
    queries = np.ascontiguousarray(queries, np.float32)
    assert out_ids.dtype == np.uint32 and out_dists.dtype == np.float32 and out_ids.flags.c_contiguous and out_dists.flags.c_contiguous
    check(lib().rg_groundtruth_rank(comm.handle, C.c_void_p(base_shard_t.data_ptr()), C.c_uint32(base_shard_t.shape[0]),
                                    C.c_uint32(base_shard_t.stride(0)), C.c_uint32(id_base), _vp(queries), C.c_uint32(queries.shape[0]),
                                    C.c_uint32(queries.shape[1]), C.c_uint32(queries.shape[1]), METRIC[metric], C.c_uint32(K),
                                    C.c_uint32(batch), _vp(out_ids), _vp(out_dists)))


def owned_rows(nq, world, rank, batch=0):
    """Row indices of an nq-query job that rank `rank` owns under rg_groundtruth_rank's batching (the rank-th balanced
    contiguous slice of every batch)."""
    qb = min(nq, batch or 65536)
    rows = []
    for q0 in range(0, nq, qb):
        n = min(qb, nq - q0)
        per, extra = divmod(n, world)
        lo = rank * per + min(rank, extra)
        rows.extend(range(q0 + lo, q0 + lo + per + (1 if rank < extra else 0)))
    return np.array(rows, np.int64)


def query_ranges(nq, world):
    """Contiguous query ranges owned by each rank for the merge step."""
    per = (nq + world - 1) // world
    return [(min(nq, r * per), min(nq, (r + 1) * per)) for r in range(world)]


def shard_rows(nb, world):
    """Balanced contiguous row shards: floor(nb/world) rows each, the first nb % world ranks one more (so that no shard
    ends up shorter than K, or empty, when nb is not a multiple of world)."""
    per, extra = divmod(nb, world)
    return [(r * per + min(r, extra), (r + 1) * per + min(r + 1, extra)) for r in range(world)]


def groundtruth_distributed(base_shard, id_base, queries, metric, K, group=None, shard_fn=None, merge_fn=None):
    """One process per GPU.  base_shard: this rank's rows (torch tensor), id_base: global id of its first row,
    queries: ALL queries (same on every rank).  Returns (ids, dists) for the query range this rank owns
    (query_ranges(nq, world)[rank]); torch tensors on base_shard's device.

    shard_fn / merge_fn default to the HIP kernels; the CPU (gloo) tests pass numpy stand-ins to exercise the
    partitioning + exchange logic without a GPU.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = base_shard.device
    nq = queries.shape[0]
    ids = torch.zeros((nq, K), dtype=torch.int32, device=dev)
    vals = torch.zeros((nq, K), dtype=torch.float32, device=dev)
    if shard_fn is None:
        gt_shard_dev(base_shard, queries, metric, K, id_base, ids, vals,
                     stream=torch.cuda.current_stream().cuda_stream)
    else:
        shard_fn(base_shard, queries, metric, K, id_base, ids, vals)
    if world == 1:
        return ids, vals
    # all-to-all: send rank j the rows of the query range j owns; pad ranges to a common length
    ranges = query_ranges(nq, world)
    per = max(hi - lo for lo, hi in ranges)
    send_i = torch.zeros((world, per, K), dtype=torch.int32, device=dev)
    send_v = torch.zeros((world, per, K), dtype=torch.float32, device=dev)
    for j, (lo, hi) in enumerate(ranges):
        send_i[j, : hi - lo] = ids[lo:hi]
        send_v[j, : hi - lo] = vals[lo:hi]
    if dist.get_backend(group) == "nccl" or not send_i.is_cuda:
        recv_i = torch.empty_like(send_i)
        recv_v = torch.empty_like(send_v)
        dist.all_to_all_single(recv_i, send_i, group=group)
        dist.all_to_all_single(recv_v, send_v, group=group)
    else:
        # gloo has no all-to-all on device tensors (control-flow tests of the multi-rank path on one GPU): pairwise
        # exchange through host memory, same result
        hi_, hv_ = send_i.cpu(), send_v.cpu()
        ri, rv = torch.empty_like(hi_), torch.empty_like(hv_)
        ri[rank], rv[rank] = hi_[rank], hv_[rank]
        ops = []
        for peer in range(world):
            if peer == rank:
                continue
            ops += [dist.P2POp(dist.isend, hi_[peer].contiguous(), peer, group), dist.P2POp(dist.isend, hv_[peer].contiguous(), peer, group),
                    dist.P2POp(dist.irecv, ri[peer], peer, group), dist.P2POp(dist.irecv, rv[peer], peer, group)]
        for w_ in dist.batch_isend_irecv(ops):
            w_.wait()
        recv_i, recv_v = ri.to(dev), rv.to(dev)
    lo, hi = ranges[rank]
    n_own = hi - lo
    out_i = torch.zeros((per, K), dtype=torch.int32, device=dev)
    out_v = torch.zeros((per, K), dtype=torch.float32, device=dev)
    if merge_fn is None:
        gt_merge_dev(recv_i, recv_v, world, per, K, metric, out_i, out_v,
                     stream=torch.cuda.current_stream().cuda_stream)
    else:
        merge_fn(recv_i, recv_v, world, per, K, metric, out_i, out_v)
    return out_i[:n_own], out_v[:n_own]
