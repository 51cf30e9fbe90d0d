"""Host-side mirror of the reference's index object for the search path.

Names and argument meaning follow efanna2e::IndexBipartite
(include/index_bipartite.h:27-138) so that tests read like the reference's
driver (tests/test_search_roargraph.cpp:160-209):

    index = IndexBipartite(dim, n, metric)
    index.LoadSearchNeededData(base_fbin)          # index_bipartite.h:62-64
    index.LoadProjectionGraph(index_file)          # index_bipartite.h:105
    index.InitVisitedListPool(T)                   # index_bipartite.h:133 (visited pool lives in HBM; no-op)
    ids, dists, cmps, hops = index.SearchRoarGraph(queries, k, L_pq)   # batch form of index_bipartite.h:100-101

Everything runs through the C ABI (include/rg.h); torch is used only to hold
device buffers when the caller wants the HBM-resident form.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import METRIC, check, lib


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class IndexBipartite:
    def __init__(self, dimension=None, n=None, metric="ip", device=0):
        self.metric = metric if isinstance(metric, str) else {0: "l2", 1: "ip", 4: "cosine"}[metric]
        self.device = device
        self.handle = C.c_void_p()
        self._base_path = None
        self._keep = None

    # ---- lifecycle -------------------------------------------------------------------------------------
    def LoadSearchNeededData(self, base_file, sampled_query_file=""):
        self._base_path = base_file

    def LoadProjectionGraph(self, index_file):
        if self._base_path is None:
            raise RuntimeError("LoadSearchNeededData must be called first")
        self.close()
        check(lib().rg_index_open(self._base_path.encode(), index_file.encode(), METRIC[self.metric], self.device,
                                  C.byref(self.handle)))

    def InitVisitedListPool(self, num_threads):
        return None

    @classmethod
    def from_arrays(cls, base, offsets, nbrs, ep, metric="ip", device=0):
        """Host numpy buffers (copied to HBM)."""
        self = cls(metric=metric, device=device)
        base = np.ascontiguousarray(base, np.float32)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        if nbrs.size == 0:
            nbrs = np.zeros(1, np.uint32)
        check(lib().rg_index_open_mem(_vp(base), C.c_uint32(base.shape[0]), C.c_uint32(base.shape[1]),
                                      C.c_uint32(base.shape[1]), _vp(offsets), _vp(nbrs), C.c_uint32(ep),
                                      METRIC[metric], device, C.byref(self.handle)))
        return self

    @classmethod
    def from_device(cls, base_t, offsets_t, nbrs_t, ep, metric="ip", dim=None):
        """torch CUDA tensors already in HBM; base is borrowed (kept alive by this object)."""
        self = cls(metric=metric, device=base_t.device.index or 0)
        import torch
        assert base_t.is_cuda and base_t.dtype == torch.float32 and base_t.is_contiguous(), "the kernels read the base as fp32 rows"
        nd, stride = base_t.shape
        check(lib().rg_index_open_dev(C.c_void_p(base_t.data_ptr()), C.c_uint32(nd), C.c_uint32(dim or stride),
                                      C.c_uint32(stride), C.c_void_p(offsets_t.data_ptr()),
                                      C.c_void_p(nbrs_t.data_ptr()), C.c_uint32(ep), METRIC[metric], self.device,
                                      C.byref(self.handle)))
        self._keep = base_t
        return self

    def close(self):
        if self.handle:
            lib().rg_index_close(self.handle)
            self.handle = C.c_void_p()

    @staticmethod
    def release_device_cache(device=0):
        """rg_mem_release: closed indexes leave their large buffers mapped in the library's cache (rg.h: rg_index_close); this hands that
        memory back to the device -- call it before another allocator of the process (torch) needs the room."""
        check(lib().rg_mem_release(C.c_int(device)))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        nd, dim, stride, ep, md, dev = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int()
        avg = C.c_float()
        check(lib().rg_index_info(self.handle, C.byref(nd), C.byref(dim), C.byref(stride), C.byref(ep), C.byref(avg),
                                  C.byref(md), C.byref(dev)))
        return dict(nd=nd.value, dim=dim.value, stride=stride.value, ep=ep.value, avg_degree=avg.value,
                    max_degree=md.value, device=dev.value)

    def set(self, name, value):
        check(lib().rg_index_set(self.handle, name.encode(), int(value)))

    def mem_stats(self):
        """rg_mem_stats of this index's device: balanced buffers made, large requests that fell back to plain allocations,
        memory classes found, 1-GiB granules of the live buffers per class"""
        nb, npl, nc = C.c_uint64(), C.c_uint64(), C.c_uint32()
        per = (C.c_uint64 * 4)()
        check(lib().rg_mem_stats(C.c_int(self.info()["device"]), C.byref(nb), C.byref(npl), C.byref(nc), per))
        ex = (C.c_uint64 * 10)()
        check(lib().rg_mem_stats_ex(C.c_int(self.info()["device"]), ex, 10))
        return {"balanced_buffers": int(nb.value), "plain_fallbacks": int(npl.value), "memory_classes_found": int(nc.value),
                "GiB_of_live_buffers_per_class": [int(x) for x in per],
                # round 5: what the placement cost and whether this index got it (rg_mem_stats_ex, rg_index_stat)
                "probe_launches": int(ex[3]), "probe_seconds": round(ex[4] / 1e6, 3), "GiB_of_address_space_reserved": round(ex[5] / 2 ** 30, 1),
                "requests_served_from_cache": int(ex[6]), "GiB_cached": round(ex[7] / 2 ** 30, 1),
                "placement_balanced": bool(self.stat("placement_balanced")), "plain_allocs_of_this_index": self.stat("plain_allocs")}

    def stat(self, name):
        """counters of the search path since open (rg_index_stat): batches_lset / batches_filter_log / batches_exact_hbm /
        batches_filter_only, lset_left, recounted"""
        v = C.c_uint64()
        check(lib().rg_index_stat(self.handle, name.encode(), C.byref(v)))
        return int(v.value)

    # ---- operator --------------------------------------------------------------------------------------
    def score_batch(self, query, ids):
        """out[i] = Distance::compare(base[ids[i]], query, dim)   (include/efanna2e/distance.h:18)"""
        query = np.ascontiguousarray(query, np.float32)
        ids = np.ascontiguousarray(ids, np.uint32)
        dim = self.info()["dim"]
        if query.shape[0] < dim:
            query = np.concatenate([query, np.zeros(dim - query.shape[0], np.float32)])
        out = np.empty(ids.shape[0], np.float32)
        check(lib().rg_score_batch(self.handle, _vp(query), _vp(ids), C.c_uint32(ids.shape[0]), _vp(out)))
        return out

    def score_batch_dev(self, query_t, ids_t, out_t, stream=0):
        check(lib().rg_score_batch_dev(self.handle, C.c_void_p(query_t.data_ptr()), C.c_void_p(ids_t.data_ptr()),
                                       C.c_uint32(ids_t.numel()), C.c_void_p(out_t.data_ptr()), C.c_void_p(stream)))

    # ---- search ----------------------------------------------------------------------------------------
    def SearchRoarGraph(self, queries, k, L_pq):
        """Batch of SearchRoarGraph calls; returns (indices[nq,k], res_dists[nq,k], cmps[nq], hops[nq])."""
        queries = np.ascontiguousarray(queries, np.float32)
        nq = queries.shape[0]
        ids = np.zeros((nq, k), np.uint32)
        dists = np.zeros((nq, k), np.float32)
        cmps = np.zeros(nq, np.uint32)
        hops = np.zeros(nq, np.uint32)
        check(lib().rg_search(self.handle, _vp(queries), C.c_uint32(nq), C.c_uint32(queries.shape[1]), C.c_uint32(k),
                              C.c_uint32(L_pq), _vp(ids), _vp(dists), _vp(cmps), _vp(hops)))
        return ids, dists, cmps, hops

    def search_dev(self, q_t, k, L_pq, ids_t, dists_t, cmps_t=None, hops_t=None, stream=0):
        """HBM-resident form: enqueue on `stream` (a raw hipStream_t value), no synchronisation."""
        check(lib().rg_search_dev(self.handle, C.c_void_p(q_t.data_ptr()), C.c_uint32(q_t.shape[0]),
                                  C.c_uint32(q_t.stride(0)), C.c_uint32(k), C.c_uint32(L_pq),
                                  C.c_void_p(ids_t.data_ptr()), C.c_void_p(dists_t.data_ptr()),
                                  C.c_void_p(cmps_t.data_ptr() if cmps_t is not None else 0),
                                  C.c_void_p(hops_t.data_ptr() if hops_t is not None else 0), C.c_void_p(stream)))

    def search_wait(self, stream=0):
        check(lib().rg_search_wait(self.handle, C.c_void_p(stream)))

    def search_prepare(self, nq, L_pq, stream=0):
        """Allocate ahead of time what batches of up to nq queries at beam widths up to L_pq need on `stream`."""
        check(lib().rg_search_prepare(self.handle, C.c_void_p(stream), C.c_uint32(nq), C.c_uint32(L_pq)))

    def reuse_stats(self, stream=0, row_counts_t=None):
        """(evaluations performed, distinct base rows among them) of the last default-mode batch on `stream`;
        row_counts_t: optional zeroed int32 CUDA tensor [nd] that receives the reads per row."""
        ev, dr = C.c_uint64(), C.c_uint64()
        check(lib().rg_search_reuse_stats(self.handle, C.c_void_p(stream), C.byref(ev), C.byref(dr),
                                          C.c_void_p(row_counts_t.data_ptr() if row_counts_t is not None else 0)))
        return ev.value, dr.value


def search_sharded(replicas, queries, k, L_pq):
    """rg_search_sharded: one IndexBipartite replica per device, the query batch split in contiguous slices that run
    concurrently; returns (indices, res_dists, cmps, hops) in query order, identical to one replica's SearchRoarGraph."""
    queries = np.ascontiguousarray(queries, np.float32)
    nq = queries.shape[0]
    ids = np.zeros((nq, k), np.uint32)
    dists = np.zeros((nq, k), np.float32)
    cmps = np.zeros(nq, np.uint32)
    hops = np.zeros(nq, np.uint32)
    handles = (C.c_void_p * len(replicas))(*[r.handle for r in replicas])
    check(lib().rg_search_sharded(handles, C.c_int(len(replicas)), _vp(queries), C.c_uint32(nq), C.c_uint32(queries.shape[1]),
                                  C.c_uint32(k), C.c_uint32(L_pq), _vp(ids), _vp(dists), _vp(cmps), _vp(hops)))
    return ids, dists, cmps, hops


# ---- formats through the C ABI -----------------------------------------------------------------------------
def fbin_meta(path):
    n, d = C.c_uint32(), C.c_uint32()
    check(lib().rg_fbin_meta(path.encode(), C.byref(n), C.byref(d)))
    return n.value, d.value


def fbin_load(path):
    n, d, s = C.c_uint32(), C.c_uint32(), C.c_uint32()
    p = C.c_void_p()
    check(lib().rg_fbin_load(path.encode(), C.byref(n), C.byref(d), C.byref(s), C.byref(p)))
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value, s.value)).copy()
    lib().rg_free(p)
    return arr, d.value


def gt_meta(path):
    n, k = C.c_uint32(), C.c_uint32()
    check(lib().rg_gt_meta(path.encode(), C.byref(n), C.byref(k)))
    return n.value, k.value


def gt_load(path):
    n, k = C.c_uint32(), C.c_uint32()
    pi, pd = C.c_void_p(), C.c_void_p()
    check(lib().rg_gt_load(path.encode(), C.byref(n), C.byref(k), C.byref(pi), C.byref(pd)))
    ids = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_uint32)), shape=(n.value, k.value)).copy()
    ds = np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_float)), shape=(n.value, k.value)).copy()
    lib().rg_free(pi)
    lib().rg_free(pd)
    return ids, ds


def gt_save(path, ids, dists):
    ids = np.ascontiguousarray(ids, np.uint32)
    dists = np.ascontiguousarray(dists, np.float32)
    check(lib().rg_gt_save(path.encode(), _vp(ids), _vp(dists), C.c_uint32(ids.shape[0]), C.c_uint32(ids.shape[1])))


def knn_ids_load(path):
    n, k = C.c_uint32(), C.c_uint32()
    p = C.c_void_p()
    check(lib().rg_knn_ids_load(path.encode(), C.byref(n), C.byref(k), C.byref(p)))
    ids = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value, k.value)).copy()
    lib().rg_free(p)
    return ids


def graph_load(path):
    nd, ep = C.c_uint32(), C.c_uint32()
    po, pn = C.c_void_p(), C.c_void_p()
    check(lib().rg_graph_load(path.encode(), C.byref(nd), C.byref(ep), C.byref(po), C.byref(pn)))
    off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(nd.value + 1,)).copy()
    ne = int(off[-1])
    nb = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint32)), shape=(max(ne, 1),)).copy()[:ne]
    lib().rg_free(po)
    lib().rg_free(pn)
    return off, nb, ep.value


def graph_save(path, offsets, nbrs, ep):
    offsets = np.ascontiguousarray(offsets, np.uint64)
    nbrs = np.ascontiguousarray(nbrs, np.uint32)
    if nbrs.size == 0:
        nbrs = np.zeros(1, np.uint32)
    check(lib().rg_graph_save(path.encode(), C.c_uint32(offsets.shape[0] - 1), C.c_uint32(ep), _vp(offsets), _vp(nbrs)))


def recall(res, gt, k):
    """ComputeRecall (tests/test_search_roargraph.cpp:23-36)"""
    res = np.ascontiguousarray(np.asarray(res)[:, :k], np.uint32)   # rg_recall reads k ids per row: recall@k of the first k
    gt = np.ascontiguousarray(gt, np.uint32)
    assert res.shape[1] == k and gt.shape[1] >= k
    return float(lib().rg_recall(C.c_uint32(res.shape[0]), C.c_uint32(k), C.c_uint32(gt.shape[1]), _vp(res), _vp(gt)))
