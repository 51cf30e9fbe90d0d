"""RoarGraph file formats in numpy (little-endian, no padding).

Used by the tests, the bench and the fixture generator to WRITE inputs; the
product reads files through the C-ABI (csrc/rg_formats.cpp), which carries the
reference's validation rules.

  .fbin   u32 npts, u32 dim, f32 data[npts][dim]             README.md:14, include/efanna2e/util.h:106-127,179-211
  .index  u32 ep, u32 npts, then per node u32 deg, u32 nbr[] src/index_bipartite.cpp:2097-2117, 2606-2619
  gt      u32 npts, u32 K, u32 ids[npts][K], f32 d[npts][K]  include/efanna2e/util.h:84-155
"""
import numpy as np


def write_fbin(path, data):
    data = np.ascontiguousarray(data, np.float32)
    with open(path, "wb") as f:
        f.write(np.array(data.shape, np.uint32).tobytes())
        f.write(data.tobytes())


def read_fbin(path):
    with open(path, "rb") as f:
        n, d = np.frombuffer(f.read(8), np.uint32)
        data = np.frombuffer(f.read(), np.float32)
    if data.size != int(n) * int(d):
        raise RuntimeError("Data file size wrong!")
    return data.reshape(int(n), int(d))


def write_gt(path, ids, dists=None):
    ids = np.ascontiguousarray(ids, np.uint32)
    with open(path, "wb") as f:
        f.write(np.array(ids.shape, np.uint32).tobytes())
        f.write(ids.tobytes())
        if dists is not None:
            f.write(np.ascontiguousarray(dists, np.float32).tobytes())


def read_gt(path):
    with open(path, "rb") as f:
        n, k = (int(x) for x in np.frombuffer(f.read(8), np.uint32))
        ids = np.frombuffer(f.read(n * k * 4), np.uint32).reshape(n, k)
        rest = f.read()
    dists = np.frombuffer(rest, np.float32).reshape(n, k) if len(rest) == n * k * 4 else None
    return ids, dists


def write_index(path, offsets, nbrs, ep):
    """CSR (offsets[nd+1], nbrs) -> .index"""
    offsets = np.asarray(offsets, np.int64)
    nbrs = np.ascontiguousarray(nbrs, np.uint32)
    nd = offsets.shape[0] - 1
    deg = (offsets[1:] - offsets[:-1]).astype(np.uint32)
    # interleave [deg_i, nbrs_i...]: position of node i's degree word = i + offsets[i]
    out = np.empty(nd + int(offsets[-1]), np.uint32)
    degpos = np.arange(nd, dtype=np.int64) + offsets[:-1]
    mask = np.ones(out.shape[0], bool)
    mask[degpos] = False
    out[degpos] = deg
    out[mask] = nbrs[: int(offsets[-1])]
    with open(path, "wb") as f:
        f.write(np.array([ep, nd], np.uint32).tobytes())
        f.write(out.tobytes())


def read_index(path):
    """.index -> (offsets u64[nd+1], nbrs u32[], ep)"""
    raw = np.fromfile(path, np.uint32)
    ep, nd = int(raw[0]), int(raw[1])
    body = raw[2:]
    offsets = np.zeros(nd + 1, np.uint64)
    # sequential walk over the degree words (vectorised in chunks is not possible: positions depend on degrees)
    pos = 0
    degs = np.empty(nd, np.uint32)
    for i in range(nd):
        dg = int(body[pos])
        degs[i] = dg
        pos += 1 + dg
    offsets[1:] = np.cumsum(degs, dtype=np.uint64)
    degpos = np.arange(nd, dtype=np.int64) + offsets[:-1].astype(np.int64)
    mask = np.ones(nd + int(offsets[-1]), bool)
    mask[degpos] = False
    nbrs = body[: nd + int(offsets[-1])][mask].copy()
    return offsets, nbrs, ep


def lists_to_csr(lists):
    deg = np.array([len(l) for l in lists], np.int64)
    offsets = np.zeros(len(lists) + 1, np.uint64)
    offsets[1:] = np.cumsum(deg)
    nbrs = np.concatenate([np.asarray(l, np.uint32) for l in lists]) if len(lists) and deg.sum() else np.zeros(0, np.uint32)
    return offsets, nbrs.astype(np.uint32)
