"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It wraps oracle/librg_oracle.so (plain-C restatement, see
rg_oracle.h) and, when present, the oracle/_ref/rg_ref binary that is compiled
from the reference's own headers.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librg_oracle.so")
REF_BIN = os.path.join(HERE, "_ref", "rg_ref")

METRIC = {"l2": 0, "ip": 1, "cosine": 4}

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.rgo_compare.restype = C.c_float
        L.rgo_compare.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint]
        L.rgo_recall.restype = C.c_float
        L.rgo_last_error.restype = C.c_char_p
        L.rgo_queue_trace.restype = C.c_size_t
        _lib = L
    return _lib


class Graph(C.Structure):
    _fields_ = [("nd", C.c_uint32), ("ep", C.c_uint32), ("offsets", C.c_void_p), ("nbrs", C.c_void_p)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def have_avx512():
    return bool(lib().rgo_have_avx512())


def use_avx512(on):
    lib().rgo_use_avx512(int(bool(on)))


def compare(metric, a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return float(lib().rgo_compare(METRIC[metric], _p(a), _p(b), a.shape[0]))


def compare_pairs(metric, a, b):
    """a, b: [n, d] -> f32[n], row-wise compare."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    n, d = a.shape
    out = np.empty(n, np.float32)
    f = lib().rgo_compare
    m = METRIC[metric]
    for i in range(n):
        out[i] = f(m, C.c_void_p(a.ctypes.data + i * d * 4), C.c_void_p(b.ctypes.data + i * d * 4), d)
    return out


def score_batch(base, metric, query, ids):
    base = np.ascontiguousarray(base, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    ids = np.ascontiguousarray(ids, np.uint32)
    out = np.empty(ids.shape[0], np.float32)
    lib().rgo_score_batch(_p(base), C.c_size_t(base.shape[1]), C.c_uint(base.shape[1]), METRIC[metric], _p(query),
                          _p(ids), C.c_size_t(ids.shape[0]), _p(out))
    return out


def normalize_rows(a):
    """In place: normalize<float> (util.h:214-225) over the rows of a C-contiguous float32 matrix."""
    assert a.dtype == np.float32 and a.flags.c_contiguous and a.ndim == 2
    lib().rgo_normalize_rows(_p(a), C.c_size_t(a.shape[0]), C.c_size_t(a.shape[1]), C.c_uint(a.shape[1]))
    return a


def projection_ep(base, dim=None):
    """CalculateProjectionep (src/index_bipartite.cpp:2004-2041)."""
    base = np.ascontiguousarray(base, np.float32)
    lib().rgo_projection_ep.restype = C.c_uint32
    return int(lib().rgo_projection_ep(_p(base), C.c_size_t(base.shape[1]), C.c_uint32(base.shape[0]),
                                       C.c_uint(dim if dim is not None else base.shape[1])))


def build_roargraph(base, knn_ids, metric, M_sq=100, M_pjbp=35, L_pjpq=500, dim=None, sched=None):
    """rgo_build_roargraph: the reference's one-thread BuildRoarGraph restated (oracle/rg_oracle_build.c).
    `sched`: phase 3 in batches (sizes summing to nb; searches of a batch see the graph as it stood when the batch began).
    Returns (offsets u64[nb+1], nbrs u32[], ep)."""
    base = np.ascontiguousarray(base, np.float32)
    knn_ids = np.ascontiguousarray(knn_ids, np.uint32)
    nb, stride = base.shape
    ep = C.c_uint32()
    po, pn = C.c_void_p(), C.c_void_p()
    sc = None if sched is None else np.ascontiguousarray(sched, np.uint32)
    if sc is not None and int(sc.sum()) != nb:
        raise ValueError("schedule does not cover the base")
    rc = lib().rgo_build_roargraph_sched(_p(base), C.c_size_t(stride), C.c_uint32(nb), C.c_uint(dim or stride), METRIC[metric], _p(knn_ids),
                                         C.c_uint32(knn_ids.shape[0]), C.c_uint32(knn_ids.shape[1]), C.c_uint32(M_sq), C.c_uint32(M_pjbp),
                                         C.c_uint32(L_pjpq), None if sc is None else _p(sc), C.c_uint32(0 if sc is None else sc.size),
                                         C.byref(ep), C.byref(po), C.byref(pn))
    if rc != 0:
        raise RuntimeError(lib().rgo_last_error().decode())
    off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(nb + 1,)).copy()
    ne = int(off[-1])
    nbrs = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint32)), shape=(max(ne, 1),)).copy()[:ne]
    lib().rgo_free(po)
    lib().rgo_free(pn)
    return off, nbrs, ep.value


def ref_projection_ep(base):
    """The same loops compiled with the reference's Release flags (oracle/_ref/rg_ref ep)."""
    base = np.ascontiguousarray(base, np.float32)
    with tempfile.TemporaryDirectory() as td:
        fin = os.path.join(td, "b.fbin")
        with open(fin, "wb") as f:
            f.write(np.array(base.shape, np.uint32).tobytes())
            f.write(base.tobytes())
        r = ref_run("ep", fin)
    for line in r.stdout.splitlines():
        if line.startswith("EP "):
            return int(line.split()[1])
    raise RuntimeError("rg_ref ep: no answer: " + r.stdout[-200:])


def queue_trace(cap, ops, ids, dists):
    ops = np.ascontiguousarray(ops, np.uint8)
    ids = np.ascontiguousarray(ids, np.uint32)
    dists = np.ascontiguousarray(dists, np.float32)
    n = ops.shape[0]
    oi = np.zeros(cap + 1, np.uint32)
    od = np.zeros(cap + 1, np.float32)
    of = np.zeros(cap + 1, np.uint8)
    pops = np.zeros(n + 1, np.uint32)
    cur = C.c_size_t(0)
    size = lib().rgo_queue_trace(C.c_size_t(cap), _p(ops), _p(ids), _p(dists), C.c_size_t(n), _p(oi), _p(od), _p(of),
                                 _p(pops), C.byref(cur))
    npop = int((ops == 1).sum())
    return dict(size=size, cur=cur.value, ids=oi[:size].copy(), dists=od[:size].copy(), flags=of[:size].copy(),
                pops=pops)


def make_graph(offsets, nbrs, ep):
    offsets = np.ascontiguousarray(offsets, np.uint64)
    nbrs = np.ascontiguousarray(nbrs, np.uint32)
    g = Graph(offsets.shape[0] - 1, ep, offsets.ctypes.data, nbrs.ctypes.data)
    g._keep = (offsets, nbrs)
    return g


def search(base, metric, offsets, nbrs, ep, queries, k, L, nthreads=1, dim=None):
    """Restated SearchRoarGraph over a batch. Returns ids[nq,k], dists[nq,k], cmps[nq], hops[nq]."""
    base = np.ascontiguousarray(base, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    g = make_graph(offsets, nbrs, ep)
    nq = queries.shape[0]
    d = dim if dim is not None else base.shape[1]
    ids = np.zeros((nq, k), np.uint32)
    dists = np.zeros((nq, k), np.float32)
    cmps = np.zeros(nq, np.uint32)
    hops = np.zeros(nq, np.uint32)
    errq = C.c_uint32(0)
    rc = lib().rgo_search(_p(base), C.c_size_t(base.shape[1]), C.c_uint(d), METRIC[metric], C.byref(g), _p(queries),
                          C.c_size_t(queries.shape[1]), C.c_uint32(nq), C.c_uint32(k), C.c_uint32(L), _p(ids),
                          _p(dists), _p(cmps), _p(hops), C.c_int(nthreads), C.byref(errq))
    if rc != 0:
        raise RuntimeError(lib().rgo_last_error().decode())
    return ids, dists, cmps, hops


def recall(res, gt, k):
    res = np.ascontiguousarray(res, np.uint32)
    gt = np.ascontiguousarray(gt, np.uint32)
    return float(lib().rgo_recall(C.c_uint32(res.shape[0]), C.c_uint32(k), C.c_uint32(gt.shape[1]), _p(res), _p(gt)))


def groundtruth_f64(base, queries, metric, K, nthreads=8, dim=None):
    base = np.ascontiguousarray(base, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    nq = queries.shape[0]
    d = dim if dim is not None else base.shape[1]
    ids = np.zeros((nq, K), np.uint32)
    dists = np.zeros((nq, K), np.float32)
    s64 = np.zeros((nq, K), np.float64)
    rc = lib().rgo_groundtruth_f64(_p(base), C.c_size_t(base.shape[1]), C.c_uint32(base.shape[0]), _p(queries),
                                   C.c_size_t(queries.shape[1]), C.c_uint32(nq), C.c_uint(d), METRIC[metric],
                                   C.c_uint32(K), _p(ids), _p(dists), _p(s64), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError(lib().rgo_last_error().decode())
    return ids, dists, s64


# ---- file formats through the oracle's loaders (error strings follow the reference) ----
def fbin_meta(path):
    n, d = C.c_uint32(), C.c_uint32()
    if lib().rgo_fbin_meta(path.encode(), C.byref(n), C.byref(d)) != 0:
        raise RuntimeError(lib().rgo_last_error().decode())
    return n.value, d.value


def fbin_load(path):
    n, d, s = C.c_uint32(), C.c_uint32(), C.c_uint32()
    ptr = C.c_void_p()
    if lib().rgo_fbin_load(path.encode(), C.byref(n), C.byref(d), C.byref(s), C.byref(ptr)) != 0:
        raise RuntimeError(lib().rgo_last_error().decode())
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n.value, s.value)).copy()
    lib().rgo_free(ptr)
    return arr, d.value


def gt_meta(path):
    n, k = C.c_uint32(), C.c_uint32()
    if lib().rgo_gt_meta(path.encode(), C.byref(n), C.byref(k)) != 0:
        raise RuntimeError(lib().rgo_last_error().decode())
    return n.value, k.value


def gt_load(path):
    n, k = C.c_uint32(), C.c_uint32()
    pi, pd = C.c_void_p(), C.c_void_p()
    if lib().rgo_gt_load(path.encode(), C.byref(n), C.byref(k), C.byref(pi), C.byref(pd)) != 0:
        raise RuntimeError(lib().rgo_last_error().decode())
    ids = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_uint32)), shape=(n.value, k.value)).copy()
    ds = np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_float)), shape=(n.value, k.value)).copy()
    lib().rgo_free(pi)
    lib().rgo_free(pd)
    return ids, ds


def index_load(path):
    g = Graph()
    if lib().rgo_index_load(path.encode(), C.byref(g)) != 0:
        raise RuntimeError(lib().rgo_last_error().decode())
    off = np.ctypeslib.as_array(C.cast(g.offsets, C.POINTER(C.c_uint64)), shape=(g.nd + 1,)).copy()
    nb = np.ctypeslib.as_array(C.cast(g.nbrs, C.POINTER(C.c_uint32)), shape=(max(int(off[-1]), 1),)).copy()[: int(off[-1])]
    ep = g.ep
    lib().rgo_graph_free(C.byref(g))
    return off, nb, ep


def index_save(path, offsets, nbrs, ep):
    g = make_graph(offsets, nbrs, ep)
    if lib().rgo_index_save(path.encode(), C.byref(g)) != 0:
        raise RuntimeError(lib().rgo_last_error().decode())


# ---- the reference-header driver (oracle/_ref/rg_ref) ----
def have_ref():
    if not os.path.exists(REF_BIN):
        return False
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return "avx512f" in flags and "avx512dq" in flags


def ref_run(*args, check=True):
    r = subprocess.run([REF_BIN, *map(str, args)], capture_output=True, text=True)
    if check and r.returncode != 0:
        raise RuntimeError("rg_ref failed: %s %s" % (r.stdout[-400:], r.stderr[-400:]))
    return r


def ref_dist(metric, a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    n, d = a.shape
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in"), os.path.join(td, "out")
        with open(fin, "wb") as f:
            f.write(np.array([n, d], np.uint32).tobytes())
            f.write(a.tobytes())
            f.write(b.tobytes())
        ref_run("dist", metric, fin, fout)
        return np.fromfile(fout, np.float32)


def ref_queue(cap, ops, ids, dists):
    ops = np.ascontiguousarray(ops, np.uint8)
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in"), os.path.join(td, "out")
        with open(fin, "wb") as f:
            f.write(np.array([cap, ops.shape[0]], np.uint32).tobytes())
            f.write(ops.tobytes())
            f.write(np.ascontiguousarray(ids, np.uint32).tobytes())
            f.write(np.ascontiguousarray(dists, np.float32).tobytes())
        ref_run("queue", fin, fout)
        raw = open(fout, "rb").read()
    size, cur, npop = np.frombuffer(raw[:12], np.uint32)
    o = 12
    rid = np.frombuffer(raw[o:o + 4 * size], np.uint32); o += 4 * size
    rd = np.frombuffer(raw[o:o + 4 * size], np.float32); o += 4 * size
    rf = np.frombuffer(raw[o:o + size], np.uint8); o += size
    pops = np.frombuffer(raw[o:o + 4 * npop], np.uint32)
    return dict(size=int(size), cur=int(cur), ids=rid, dists=rd, flags=rf, pops=pops)


def ref_search(base_fbin, index_path, query_fbin, metric, k, L, threads=1, repeat=1, prefetch=True):
    """oracle/_ref/rg_ref search; prefetch=True issues the reference's software prefetches (index_bipartite.cpp:2324, 2374-2375)."""
    with tempfile.TemporaryDirectory() as td:
        fout = os.path.join(td, "out")
        r = ref_run("search", base_fbin, index_path, query_fbin, metric, k, L, threads, fout, repeat, 1 if prefetch else 0, check=False)
        if r.returncode != 0:
            raise RuntimeError((r.stdout + r.stderr).strip().splitlines()[-1])
        raw = open(fout, "rb").read()
    nq, kk = np.frombuffer(raw[:8], np.uint32)
    o = 8
    ids = np.frombuffer(raw[o:o + 4 * nq * kk], np.uint32).reshape(nq, kk); o += 4 * nq * kk
    ds = np.frombuffer(raw[o:o + 4 * nq * kk], np.float32).reshape(nq, kk); o += 4 * nq * kk
    cmps = np.frombuffer(raw[o:o + 4 * nq], np.uint32); o += 4 * nq
    hops = np.frombuffer(raw[o:o + 4 * nq], np.uint32)
    qps = None
    for line in r.stdout.splitlines():
        if line.startswith("QPS "):
            qps = float(line.split()[1])
    return ids, ds, cmps, hops, qps


PRUNE_KINDS = {"get_base": 0, "reverse": 1, "reverse_phantoms": 2, "search": 3}


def prune(base, metric, M, kind, pivot, ids, dists=None, have=None):
    """rgo_prune: one call of one pruning rule of the restated construction (oracle/rg_oracle_build.c)."""
    base = np.ascontiguousarray(base, np.float32)
    ids = np.ascontiguousarray(ids, np.uint32)
    dists = np.ascontiguousarray(dists if dists is not None else np.zeros(ids.size), np.float32)
    have = np.ascontiguousarray(have if have is not None else np.zeros(0), np.uint32)
    out = np.zeros(max(int(M), ids.size) + 1, np.uint32)
    n = C.c_uint32()
    lib().rgo_prune(_p(base), C.c_size_t(base.shape[1]), C.c_uint32(base.shape[0]), C.c_uint(base.shape[1]), METRIC[metric], C.c_uint32(M),
                    C.c_int(PRUNE_KINDS[kind]), C.c_uint32(pivot), _p(ids), _p(dists), C.c_uint32(ids.size), _p(have), C.c_uint32(have.size),
                    _p(out), C.byref(n))
    return out[: n.value].copy()


def prune_calls_pack(calls):
    """calls = [(kind, pivot, ids, dists or None, have or None)] -> the bytes `rg_ref prune` reads"""
    parts = [np.array([len(calls)], np.uint32).tobytes()]
    for kind, pivot, ids, dists, have in calls:
        ids = np.ascontiguousarray(ids, np.uint32)
        dists = np.ascontiguousarray(dists if dists is not None else np.zeros(ids.size), np.float32)
        have = np.ascontiguousarray(have if have is not None else np.zeros(0), np.uint32)
        parts += [np.array([PRUNE_KINDS[kind], pivot, ids.size, have.size], np.uint32).tobytes(), ids.tobytes(), dists.tobytes(), have.tobytes()]
    return b"".join(parts)


def ref_prune(base_fbin, metric, M, calls):
    """The same calls through oracle/_ref/rg_ref prune (the reference's own Distance / Neighbor objects); list of id arrays."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        cin, cout = os.path.join(td, "calls.bin"), os.path.join(td, "out.bin")
        open(cin, "wb").write(prune_calls_pack(calls))
        ref_run("prune", base_fbin, metric, str(M), cin, cout)
        w = np.fromfile(cout, np.uint32)
    res, pos = [], 0
    for _ in calls:
        n = int(w[pos]); res.append(w[pos + 1: pos + 1 + n].copy()); pos += 1 + n
    assert pos == w.size
    return res
