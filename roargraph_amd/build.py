"""Graph construction through the C ABI (CPU code, like the reference's BuildRoarGraph)."""
import ctypes as C

import numpy as np

from ._lib import METRIC, check, lib


def build_roargraph(base, knn_ids, metric, M_sq=100, M_pjbp=35, L_pjpq=500, num_threads=1, dim=None, device=None,
                    batch=0):
    """base [nb, stride] fp32, knn_ids [nq, K] (train-query ground truth, best first).
    Returns (offsets u64[nb+1], nbrs u32[], ep).  Defaults are the paper's parameters (README.md:92-97).
    device=None: all on the CPU (one thread = the reference's T=1 result; more threads = one result for any thread count,
    phase 3 in the batches of build_schedule(nb)); device=N: the searches and prunings on GPU N (same result for batch=0)."""
    base = np.ascontiguousarray(base, np.float32)
    knn_ids = np.ascontiguousarray(knn_ids, np.uint32)
    nb, stride = base.shape
    ep = C.c_uint32()
    po, pn = C.c_void_p(), C.c_void_p()
    common = (base.ctypes.data_as(C.c_void_p), C.c_uint32(nb), C.c_uint32(dim or stride), C.c_uint32(stride),
              knn_ids.ctypes.data_as(C.c_void_p), C.c_uint32(knn_ids.shape[0]), C.c_uint32(knn_ids.shape[1]), METRIC[metric],
              C.c_uint32(M_sq), C.c_uint32(M_pjbp), C.c_uint32(L_pjpq), C.c_uint32(num_threads))
    if device is None:
        check(lib().rg_build_roargraph(*common, C.byref(ep), C.byref(po), C.byref(pn)))
    else:
        check(lib().rg_build_roargraph_gpu(*common, C.c_int(device), C.c_uint32(batch), C.byref(ep), C.byref(po), C.byref(pn)))
    off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(nb + 1,)).copy()
    ne = int(off[-1])
    nbrs = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint32)), shape=(max(ne, 1),)).copy()[:ne]
    lib().rg_free(po)
    lib().rg_free(pn)
    return off, nbrs, ep.value


def build_schedule(nb, batch=0):
    """rg_build_schedule: the batch sizes phase 3 runs in (what the oracle's scheduled build takes)."""
    n = C.c_uint32()
    check(lib().rg_build_schedule(C.c_uint32(nb), C.c_uint32(batch), None, C.c_uint32(0), C.byref(n)))
    out = np.zeros(max(n.value, 1), np.uint32)
    check(lib().rg_build_schedule(C.c_uint32(nb), C.c_uint32(batch), out.ctypes.data_as(C.c_void_p), C.c_uint32(out.size), C.byref(n)))
    return out[: n.value]


PRUNE_KINDS = {"get_base": 0, "reverse": 1, "reverse_phantoms": 2, "search": 3}


def prune_debug(base, metric, M, kind, pivot, ids, dists=None, have=None, use_gpu=False, device=0):
    """rg_build_prune_debug: ONE call of one occlusion-pruning rule of the construction (index_bipartite.cpp:1434-1694, 1846-1940) through
    the builder's host routine (use_gpu=False: no GPU needed) or the pruning kernel of the GPU-assisted build (kinds get_base / search)."""
    import ctypes as C
    import numpy as np
    from ._lib import METRIC, check, lib
    base = np.ascontiguousarray(base, np.float32)
    ids = np.ascontiguousarray(ids, np.uint32)
    dists = np.ascontiguousarray(dists if dists is not None else np.zeros(ids.size), np.float32)
    have = np.ascontiguousarray(have if have is not None else np.zeros(0), np.uint32)
    out = np.zeros(max(int(M), ids.size) + 1, np.uint32)
    n = C.c_uint32()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().rg_build_prune_debug(vp(base), C.c_uint32(base.shape[0]), C.c_uint32(base.shape[1]), C.c_uint32(base.shape[1]), METRIC[metric], C.c_uint32(M),
                                     C.c_int(kind if isinstance(kind, int) else PRUNE_KINDS[kind]), C.c_uint32(pivot), vp(ids), vp(dists), C.c_uint32(ids.size),
                                     vp(have), C.c_uint32(have.size), vp(out), C.byref(n), C.c_int(1 if use_gpu else 0), C.c_int(device)))
    return out[: n.value].copy()
