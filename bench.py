#!/usr/bin/env python3
"""bench.py -- QPS of the RoarGraph search hot path on MI355X (BASELINE.json metric #1).

One "step" = one pass of the hot path over one batch: SearchRoarGraph for a batch of 10,000 queries
(top-10, L_pq = 500) against a 10M x 200 inner-product base, the configuration of BASELINE.json configs[1]
("t2i-10M d=200 IP ... 1xMI355X search kernel").  Inputs are synthetic (no dataset can be downloaded here):
base ~ N(0,1), queries ~ N(0.3, 0.5^2), and by default a random out-degree-40 graph, which drives the same HBM
access pattern as a real index (one random 800-byte row per distance evaluation) but has no meaningful recall.
`--real-index` builds a genuine RoarGraph index for the same base inside the run (K2 ground truth + GPU-assisted
build, about 5 minutes at 10M -- too long for the default run); a smaller genuine index is always built for the
recall check (`recall_check_roargraph_index`).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the index is replicated, every rank searches its own batch of queries (independent units, no
data-path collective); value = total queries / max-over-ranks time; scaling = weak.

The JSON line also carries
  roofline      achieved = algorithmic bytes (sum of distance evaluations x 4*dim) / kernel time (HIP events on
                the launch stream), against the 8 TB/s HBM peak
  cpu_baseline  the same workload on this box's host cores (rank 0, N=1 only), bounded sample:
                kind "reference" = oracle/_ref/rg_ref (the reference's own distance/queue/visited code),
                kind "port" = oracle/librg_oracle.so (AVX-512 restatement) when the former cannot run here.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to "
                    "exercise the multi-rank control flow on a box with one GPU)")
    ap.add_argument("--nb", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=200)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--L", type=int, default=500)
    ap.add_argument("--deg", type=int, default=40)
    ap.add_argument("--metric", default="ip")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--visited", type=int, default=2,
                    help="2 = LDS visited filter + id log + exact distinct count (library default; ids, dists, hops and "
                         "cmps bit-exact); 1 = LDS filter only (cmps = evaluations performed); 0 = visited words in HBM")
    ap.add_argument("--filter-log2", type=int, default=0, help="LDS visited-filter size (log2 entries); 0 = the library's automatic choice")
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--rows-per-pass", type=int, default=0)
    ap.add_argument("--no-other-modes", action="store_true", help="skip timing the non-default visited modes (profiling runs)")
    ap.add_argument("--sweep", default="", help="comma list of extra L_pq values to report (not part of the timed metric)")
    ap.add_argument("--gt-nq", type=int, default=65536, help="queries of the ground-truth (K2) leg; 0 = skip")
    ap.add_argument("--gt-K", type=int, default=100)
    ap.add_argument("--real-index", action="store_true",
                    help="build a genuine RoarGraph index for the bench base inside the run (K2 ground truth of --train "
                         "queries, GPU-assisted build; about 5 min at 10M) instead of the random graph")
    ap.add_argument("--data", default="gaussian", help="gaussian (default, hardest: no structure) | lowrank (structured embeddings)")
    ap.add_argument("--rank", type=int, default=32, help="latent rank of --data lowrank")
    ap.add_argument("--train", type=int, default=0, help="training queries for --real-index (default nb/5)")
    ap.add_argument("--recall-nb", type=int, default=200_000,
                    help="base size of the recall check on a genuine RoarGraph index built in the run; 0 = skip")
    return ap.parse_args()


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(args, base_t, off_t, nbrs_t, ep, q_t, ids_gpu, budget_s):
    """Time the CPU path on a bounded sample of the same workload; also re-checks parity on that sample."""
    from oracle import pyoracle as po
    from roargraph_amd import io
    po.build() if not os.path.exists(po.LIB_PATH) else None
    ncores = os.cpu_count() or 1
    threads = min(16, ncores)  # README.md:110 evaluates with 16 threads
    base = base_t.cpu().numpy()
    off = off_t.cpu().numpy().view(np.uint64)
    nbrs = nbrs_t.cpu().numpy().view(np.uint32)
    q = q_t.cpu().numpy()
    out = {"unit": "QPS", "cores": threads, "host_cores": ncores}
    # pilot with the C port to size the sample
    po.use_avx512(True)
    pilot = min(args.nq, 2 * threads)
    t0 = time.time()
    r = po.search(base, args.metric, off, nbrs, ep, q[:pilot], args.k, args.L, nthreads=threads)
    dt = max(time.time() - t0, 1e-6)
    assert (r[0] == ids_gpu[:pilot]).all(), "CPU oracle and GPU disagree on the bench workload"
    n = int(min(args.nq, max(pilot, budget_s * pilot / dt)))
    n = max(threads, n - n % threads)
    if po.have_ref() and _mem_available_gb() > 6.0 * base.nbytes / 1e9:
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            bf, qf, gf = (os.path.join(td, x) for x in ("b.fbin", "q.fbin", "g.index"))
            io.write_fbin(bf, base)
            io.write_fbin(qf, q[:n])
            io.write_index(gf, off, nbrs, ep)
            ids, _, cmps, _, qps = po.ref_search(bf, gf, qf, args.metric, args.k, args.L, threads=threads)
        assert (ids == ids_gpu[:n]).all(), "reference-header driver and GPU disagree on the bench workload"
        out.update(value=qps, kind="reference",
                   sample="%d of %d queries, %d OpenMP threads, oracle/_ref/rg_ref (reference distance.h/neighbor.h/"
                          "visited_list_pool.h, search loop restated)" % (n, args.nq, threads))
    else:
        t0 = time.time()
        r = po.search(base, args.metric, off, nbrs, ep, q[:n], args.k, args.L, nthreads=threads)
        dt = time.time() - t0
        assert (r[0] == ids_gpu[:n]).all()
        out.update(value=n / dt, kind="port",
                   sample="%d of %d queries, %d OpenMP threads, oracle/librg_oracle.so (avx512=%s)"
                          % (n, args.nq, threads, bool(po.have_avx512())))
    out["mean_evals"] = float(np.mean(cmps if out["kind"] == "reference" else r[2]))
    return out


def recall_check(args, dev, structured=False):
    """QPS @ recall@10 on a GENUINE RoarGraph index (rank 0, N=1), built here with the reference's own pipeline and
    parameters on synthetic cross-modal data: K2 ground truth of the training queries -> rg_build_roargraph on the host
    cores (M_sq=100, M_pjbp=35, L_pjpq=500, README.md:92-97) -> K1 search, K2 truth for the test queries.
    The 10M bench graph is random (a 10M-node build takes far longer than a bench run), so this smaller set is where
    recall is measured; profiles/r01/e2e_pipeline_1m.json holds the same pipeline at 1M x 200."""
    import torch
    from roargraph_amd import build, groundtruth, index
    from roargraph_amd.index import IndexBipartite
    nb, ntrain, nq, dim, metric = args.recall_nb, args.recall_nb // 2, 5000, args.dim, args.metric
    if structured:
        # the second check: embeddings with a low intrinsic dimension (where a graph index reaches high recall), five times
        # the rows, phase 3 of the build on the GPU
        from roargraph_amd import synth
        nb, ntrain = 5 * args.recall_nb, args.recall_nb
        base, train, q, desc = synth.make_device_set(dev, 1234, nb, ntrain, nq, dim, data="lowrank", rank=args.rank)
    else:
        g = torch.Generator(device=dev); g.manual_seed(1234)
        base = torch.empty((nb, dim), device=dev).normal_(generator=g)
        train = torch.empty((ntrain, dim), device=dev).normal_(generator=g) * 0.5 + 0.3
        q = torch.empty((nq, dim), device=dev).normal_(generator=g) * 0.5 + 0.3
        desc = "base N(0,1) %d x %d, %d train / %d test queries N(0.3,0.5^2)" % (nb, dim, ntrain, nq)
    st = torch.cuda.current_stream().cuda_stream
    ti = torch.zeros((ntrain, 100), dtype=torch.int32, device=dev); tv = torch.zeros((ntrain, 100), device=dev)
    groundtruth.gt_shard_dev(base, train, metric, 100, 0, ti, tv, stream=st); torch.cuda.synchronize()
    threads = min(128, os.cpu_count() or 1)
    t0 = time.perf_counter()
    off, nbrs, ep = build.build_roargraph(base.cpu().numpy(), ti.cpu().numpy().view(np.uint32), metric, 100, 35, 500,
                                          num_threads=threads, device=(dev.index or 0) if structured else None)
    t_build = time.perf_counter() - t0
    deg = np.diff(off.astype(np.int64))
    gi = torch.zeros((nq, 100), dtype=torch.int32, device=dev); gv = torch.zeros((nq, 100), device=dev)
    groundtruth.gt_shard_dev(base, q, metric, 100, 0, gi, gv, stream=st); torch.cuda.synchronize()
    gt = gi.cpu().numpy().view(np.uint32)
    ix = IndexBipartite.from_device(base, torch.from_numpy(off.view(np.int64)).to(dev),
                                    torch.from_numpy(nbrs.view(np.int32)).to(dev), ep, metric=metric)
    ids = torch.zeros((nq, 10), dtype=torch.int32, device=dev); ds = torch.zeros((nq, 10), device=dev)
    cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
    rows = []
    for L in (50, 100, 200, 500, 1000):
        ix.search_dev(q, 10, L, ids, ds, cm, hp, stream=st); ix.search_wait(st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ix.search_dev(q, 10, L, ids, ds, cm, hp, stream=st); b.record(); ix.search_wait(st)
        ms = a.elapsed_time(b)
        rows.append({"L_pq": L, "qps": nq / (ms / 1e3), "recall_at_10": index.recall(ids.cpu().numpy().view(np.uint32), gt, 10),
                     "mean_evals": float(cm.float().mean()), "mean_hops": float(hp.float().mean())})
    ix.close()
    return {"dataset": "%s, %s" % (desc, metric),
            "index": "RoarGraph built by %s (M_sq=100, M_pjbp=35, L_pjpq=500) on %d host threads in %.1f s; "
                     "degree avg %.1f max %d" % ("rg_build_roargraph_gpu" if structured else "rg_build_roargraph", threads, t_build,
                                                 deg.mean(), deg.max()),
            "queries": nq, "curve": rows}


def gt_cpu_baseline(base, gq, args):
    """CPU baseline of the ground-truth leg: oracle/gt_numpy.py (blocked SGEMM on all host cores + per-query top-K, the
    shape of the reference's compute_groundtruth) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle"))
    import gt_numpy
    nbs, nqs = min(args.nb, 1_000_000), min(args.gt_nq, 2048)
    hb = base[:nbs].cpu().numpy()
    hq = gq[:nqs].cpu().numpy()
    gt_numpy.groundtruth_blocked(hb[:65536], hq[:64], args.metric, args.gt_K)   # warm the BLAS threads
    t0 = time.perf_counter()
    gt_numpy.groundtruth_blocked(hb, hq, args.metric, args.gt_K)
    dt = time.perf_counter() - t0
    return {"value": float(nbs) * float(nqs) / dt, "unit": "distances/s", "cores": os.cpu_count() or 1, "kind": "port",
            "sample": "%d queries x %d base rows, K=%d, numpy/OpenBLAS SGEMM + argpartition per 131072-row block "
                      "(oracle/gt_numpy.py), %.1f s" % (nqs, nbs, args.gt_K, dt)}


def pmc_traffic(args):
    """HBM bytes per launch of the search kernel from the committed rocprofv3 PMC passes (profiles/*/search_traffic.json,
    written by scripts/profile_on_box.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE
    with the gfx950 x2 correction).  Counters cannot be read from inside the timed process, so the figure is only
    reported when the profiled workload is the one being benched; otherwise null."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    key = {"nb": args.nb, "dim": args.dim, "nq": args.nq, "L": args.L, "k": args.k, "deg": args.deg,
           "metric": args.metric, "visited": args.visited, "real_index": bool(args.real_index)}
    if args.data != "gaussian":
        return None, None
    for path in sorted(glob.glob(os.path.join(here, "profiles", "*", "search_traffic.json")), reverse=True):
        try:
            t = json.load(open(path))
        except Exception:
            continue
        if t.get("workload") == key and t.get("fetch_bytes_corrected"):
            total = float(t["fetch_bytes_corrected"]) + float(t.get("write_bytes") or 0.0)
            return total, "%s (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction %.4g B + WRITE_SIZE %.4g B per launch)" % (
                os.path.relpath(path, here), t["fetch_bytes_corrected"], t.get("write_bytes") or 0.0)
    return None, None


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from roargraph_amd._lib import lib
    from roargraph_amd.index import IndexBipartite

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if lib().rg_device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local = local % torch.cuda.device_count()   # (control-flow tests run several ranks on one GPU with --backend gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    # ---- synthetic t2i-10M-shaped inputs, resident in HBM before the timed region -------------------------------
    g = torch.Generator(device=dev)
    g.manual_seed(1234)  # same base + graph on every rank (replicated index)
    ntrain = (args.train or args.nb // 5) if args.real_index else 0
    lowrank = None
    if args.data == "lowrank":
        # embeddings with a low intrinsic dimension (the set the recall target is demonstrated on); rank 0 data on every rank
        from roargraph_amd import synth
        base, train, lowrank_q, data_desc = synth.make_device_set(dev, 1234, args.nb, ntrain, args.nq, args.dim, data="lowrank",
                                                                  rank=args.rank)
        lowrank = lowrank_q
    else:
        base = torch.empty((args.nb, args.dim), dtype=torch.float32, device=dev)
        chunk = 1 << 20
        for s in range(0, args.nb, chunk):
            base[s:s + chunk].normal_(generator=g)
        data_desc = "synthetic N(0,1) base"
    graph_desc = "random out-degree-%d graph" % args.deg
    if args.real_index:
        from roargraph_amd import build, groundtruth
        if lowrank is None:
            train = torch.empty((ntrain, args.dim), dtype=torch.float32, device=dev).normal_(generator=g) * 0.5 + 0.3
        ti = torch.zeros((ntrain, 100), dtype=torch.int32, device=dev); tv = torch.zeros((ntrain, 100), device=dev)
        t0 = time.perf_counter()
        groundtruth.gt_shard_dev(base, train, args.metric, 100, 0, ti, tv, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        t_gt = time.perf_counter() - t0
        t0 = time.perf_counter()
        h_off, h_nbrs, ep = build.build_roargraph(base.cpu().numpy(), ti.cpu().numpy().view(np.uint32), args.metric, 100, 35, 500,
                                                  num_threads=min(128, os.cpu_count() or 1), device=local)
        t_build = time.perf_counter() - t0
        off = torch.from_numpy(h_off.view(np.int64)).to(dev)
        nbrs = torch.from_numpy(h_nbrs.view(np.int32)).to(dev)
        graph_desc = ("genuine RoarGraph index built in the run (K2 truth of %d training queries %.0f s, GPU-assisted build "
                      "%.0f s, M_sq=100 M_pjbp=35 L_pjpq=500, avg degree %.1f)" % (ntrain, t_gt, t_build, h_nbrs.size / args.nb))
        del train, ti, tv
    else:
        nbrs = torch.randint(0, args.nb, (args.nb * args.deg,), dtype=torch.int32, device=dev, generator=g)
        off = torch.arange(0, args.nb + 1, dtype=torch.int64, device=dev) * args.deg
        ep = 0
    g.manual_seed(99 + rank)  # each rank searches its own query batch
    if lowrank is not None:
        q = lowrank      # same query batch on every rank for this data set
    else:
        q = torch.empty((args.nq, args.dim), dtype=torch.float32, device=dev).normal_(generator=g) * 0.5 + 0.3
    ids = torch.zeros((args.nq, args.k), dtype=torch.int32, device=dev)
    dists = torch.zeros((args.nq, args.k), dtype=torch.float32, device=dev)
    cmps = torch.zeros(args.nq, dtype=torch.int32, device=dev)
    hops = torch.zeros(args.nq, dtype=torch.int32, device=dev)
    index = IndexBipartite.from_device(base, off, nbrs, ep, metric=args.metric)
    if args.waves_per_cu:
        index.set("waves_per_cu", args.waves_per_cu)
    if args.rows_per_pass:
        index.set("rows_per_pass", args.rows_per_pass)
    stream = torch.cuda.current_stream().cuda_stream
    if args.filter_log2:
        index.set("filter_log2", args.filter_log2)

    # reference-equivalent evaluation counts (exact visited mode), and a parity check between the two modes
    index.set("visited", 0)
    index.search_dev(q, args.k, args.L, ids, dists, cmps, hops, stream=stream)
    index.search_wait(stream)
    ref_ids, ref_dists, ref_cmps, ref_hops = ids.clone(), dists.clone(), cmps.clone(), hops.clone()
    index.set("visited", args.visited)

    def step(L):
        index.search_dev(q, args.k, L, ids, dists, cmps, hops, stream=stream)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(args.L)
    index.search_wait(stream)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync_all()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step(args.L)
        b.record()
    sync_all()
    t1 = time.perf_counter()
    index.search_wait(stream)
    elapsed = t1 - t0
    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_q = args.nq * args.steps * world
    qps = total_q / elapsed
    assert torch.equal(ids, ref_ids) and torch.equal(hops, ref_hops), "visited modes disagree on ids/hops"
    assert torch.equal(dists.view(torch.int32), ref_dists.view(torch.int32)), "visited modes disagree on distances"
    assert bool((cmps >= ref_cmps).all())
    if args.visited != 1:
        assert torch.equal(cmps, ref_cmps), "cmps differ from the exact visited mode"
    mean_cmps = float(ref_cmps.float().mean().item())      # the reference's avg_visited (distinct nodes scored)
    mean_done = float(cmps.float().mean().item())           # evaluations this mode actually performed
    mean_hops = float(hops.float().mean().item())
    kavg = float(np.mean(kernel_ms)) / 1e3
    # algorithmic bytes: the REFERENCE's evaluation count x 4*dim (re-scored repeats of the filter mode are not credited)
    alg_bytes = float(ref_cmps.to(torch.int64).sum().item()) * 4.0 * args.dim
    achieved = alg_bytes / kavg / 1e9

    # recall@10 of the timed search against exact truth from K2 (meaningless on the random graph, real on --real-index)
    from roargraph_amd import groundtruth as _gtmod, index as _ixmod
    ti_q = torch.zeros((args.nq, 100), dtype=torch.int32, device=dev); tv_q = torch.zeros((args.nq, 100), device=dev)
    _gtmod.gt_shard_dev(base, q, args.metric, 100, 0, ti_q, tv_q, stream=stream); torch.cuda.synchronize()
    gt_np = ti_q.cpu().numpy().view(np.uint32)
    recall10 = _ixmod.recall(ids.cpu().numpy().view(np.uint32), gt_np, 10) if args.k >= 10 else None
    del ti_q, tv_q

    other = None
    if rank == 0 and not args.no_other_modes:
        other = []
        for om in (0, 1, 2):
            if om == args.visited:
                continue
            index.set("visited", om)
            step(args.L); torch.cuda.synchronize()
            oe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(3, args.steps))]
            for a, b in oe:
                a.record(); step(args.L); b.record()
            torch.cuda.synchronize()
            oms = float(np.mean([a.elapsed_time(b) for a, b in oe]))
            other.append({"visited": om, "qps": args.nq / (oms / 1e3), "ms_avg": oms, "achieved_GBps": alg_bytes / (oms / 1e3) / 1e9})
        index.set("visited", args.visited)

    # opt-in NON-parity fast mode (rg_index_set "fast_bf16"): reported separately, never as `value`
    fast = None
    if rank == 0 and not args.no_other_modes and args.dim in (200, 512):
        try:
            index.set("fast_bf16", 1)
            f_ids = torch.zeros_like(ids); f_d = torch.zeros_like(dists); f_c = torch.zeros_like(cmps); f_h = torch.zeros_like(hops)
            index.search_dev(q, args.k, args.L, f_ids, f_d, f_c, f_h, stream=stream); index.search_wait(stream)
            fe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(3, args.steps))]
            for a, b in fe:
                a.record(); index.search_dev(q, args.k, args.L, f_ids, f_d, f_c, f_h, stream=stream); b.record()
            torch.cuda.synchronize(); index.search_wait(stream)
            fms = float(np.mean([a.elapsed_time(b) for a, b in fe]))
            fi, ei = f_ids.cpu().numpy().view(np.uint32), ref_ids.cpu().numpy().view(np.uint32)
            same = float(np.mean([len(set(fi[i].tolist()) & set(ei[i].tolist())) / float(args.k) for i in range(args.nq)]))
            row_b = (args.dim + 127) // 128 * 256
            fast = {"mode": "fast_bf16 (opt-in, not parity: bf16 traversal + exact fp32 re-rank of the beam)",
                    "qps": args.nq / (fms / 1e3), "ms_avg": fms, "overlap_with_exact_search_top%d" % args.k: same,
                    "recall_at_10": _ixmod.recall(fi, gt_np, 10) if args.k >= 10 else None,
                    "mean_evals_performed": float(f_c.float().mean().item()),
                    "hbm_bytes_per_evaluation": row_b,
                    "row_GBps": float(f_c.to(torch.int64).sum().item()) * row_b / (fms / 1e3) / 1e9}
        except Exception as e:
            fast = {"error": repr(e)}
        index.set("fast_bf16", 0)
        step(args.L); torch.cuda.synchronize(); index.search_wait(stream)

    sweep = []
    if args.sweep and rank == 0:
        for L in [int(x) for x in args.sweep.split(",")]:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(2):                        # (the wait is where the default mode adapts: one batch to measure
                step(L); index.search_wait(stream)    #  the re-scoring ratio, one for the timed trial of the exact words)
            a.record(); step(L); b.record(); torch.cuda.synchronize(); index.search_wait(stream)
            ms = a.elapsed_time(b)
            mc = float(cmps.float().mean().item())
            row = {"L_pq": L, "qps": args.nq / (ms / 1e3), "mean_evals": mc,
                   "recall_at_10": _ixmod.recall(ids.cpu().numpy().view(np.uint32), gt_np, 10) if args.k >= 10 else None,
                   "gbps": args.nq * mc * 4 * args.dim / (ms / 1e3) / 1e9}
            if fast is not None and "error" not in fast:   # the opt-in non-parity mode at the same L_pq, reported beside it
                index.set("fast_bf16", 1)
                step(L); torch.cuda.synchronize()
                a.record(); step(L); b.record(); torch.cuda.synchronize()
                row["fast_bf16_qps"] = args.nq / (a.elapsed_time(b) / 1e3)
                row["fast_bf16_recall_at_10"] = _ixmod.recall(ids.cpu().numpy().view(np.uint32), gt_np, 10) if args.k >= 10 else None
                index.set("fast_bf16", 0)
            sweep.append(row)
        step(args.L); torch.cuda.synchronize()

    # ---- second BASELINE metric: ground-truth build, distances/s.  Base rows sharded over the ranks (each rank scores
    # ALL gt queries against its rows), per-shard top-K exchanged with one all-to-all over RCCL, merged by K3.
    gt = None
    if args.gt_nq > 0:
        from roargraph_amd import groundtruth
        lo, hi = groundtruth.shard_rows(args.nb, world)[rank]
        g.manual_seed(4242)
        gq = torch.empty((args.gt_nq, args.dim), dtype=torch.float32, device=dev).normal_(generator=g) * 0.5 + 0.3
        shard = base[lo:hi]
        groundtruth.groundtruth_distributed(shard[: min(hi - lo, 65536)], lo, gq[:2048], args.metric, args.gt_K)  # warm-up
        sync_all()
        tg0 = time.perf_counter()
        gi, gv = groundtruth.groundtruth_distributed(shard, lo, gq, args.metric, args.gt_K)
        sync_all()
        tg = time.perf_counter() - tg0
        if world > 1:
            t = torch.tensor([tg], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tg = float(t.item())
        dps = float(args.gt_nq) * float(args.nb) / tg
        gt = {"metric": "GT-build distances/sec (K=%d, %d queries x %d base rows, base sharded x%d)" % (args.gt_K, args.gt_nq, args.nb, world),
              "value": dps, "seconds": tg, "TFLOPs_fp32_mfma": 2.0 * args.dim * dps / 1e12,
              "roofline": {"bound": "mfma", "achieved": 2.0 * args.dim * dps / 1e12, "peak": 157.3 * world, "unit": "TFLOP/s",
                           "frac": 2.0 * args.dim * dps / 1e12 / (157.3 * world)}}
        if rank == 0 and world == 1 and args.cpu_seconds > 0:
            try:
                gt["cpu_baseline"] = gt_cpu_baseline(base, gq, args)
            except Exception as e:
                gt["cpu_baseline"] = {"value": None, "unit": "distances/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        del gi, gv, gq

    rcheck = rcheck2 = None
    if rank == 0 and world == 1 and args.recall_nb > 0:
        try:
            rcheck = recall_check(args, dev)
        except Exception as e:
            rcheck = {"error": repr(e)}
        try:
            rcheck2 = recall_check(args, dev, structured=True)
        except Exception as e:
            rcheck2 = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        try:
            cpu = cpu_baseline(args, base, off, nbrs, ep, q, ids.cpu().numpy().view(np.uint32), args.cpu_seconds)
        except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
            cpu = {"value": None, "unit": "QPS", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}

    traffic, traffic_src = pmc_traffic(args) if rank == 0 else (None, None)
    if rank == 0:
        line = {
            "metric": "QPS @ recall@10, t2i-10M d=200 IP (search, top-%d, L_pq=%d)" % (args.k, args.L),
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "t2i-10M-shaped: base %dx%d fp32 %s, %d queries/GPU/step, top-%d, L_pq=%d, "
                                   "%s, %s (replicated per GPU)"
                                   % (args.nb, args.dim, args.metric, args.nq, args.k, args.L, data_desc, graph_desc),
                       "parallelism": "query-sharded x%d, index replicated" % world,
                       "recall_at_10": recall10,
                       "recall_note": ("recall of the timed search on the genuine index" if args.real_index else
                                       "random graph: same HBM access pattern as a real index, recall not meaningful (the value "
                                       "above is what it is); recall IS meaningful on the genuine RoarGraph index built in this run "
                                       "(smaller bases) in recall_check_roargraph_index / recall_check_structured_data, and with --real-index"),
                       "visited": {2: "default: lds-filter (%s) + id log + exact distinct count, adaptive to the exact HBM words where "
                                      "a timed trial finds them faster (ids/dists/hops/cmps bit-exact vs the HBM-visited mode, "
                                      "checked in this run)" % ("2^%d" % args.filter_log2 if args.filter_log2 else "auto size"),
                                   1: "lds-filter (%s) only (ids/dists/hops bit-exact; cmps = evaluations performed)" % ("2^%d" % args.filter_log2 if args.filter_log2 else "auto size"),
                                   0: "exact visited words in HBM"}[args.visited],
                       "mean_evals_per_query": mean_cmps, "mean_evals_performed": mean_done, "mean_hops": mean_hops},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "rg_search_kernel (+ rg_distinct_kernel in visited mode 2)", "kernel_ms_avg": kavg * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "frac_of_measured_stream_ceiling_6290": achieved / 6290.0},
            "cpu_baseline": cpu,
            "other_visited_modes": other,
            "fast_mode_bf16": fast,
            "gt_build": gt,
            "recall_check_roargraph_index": rcheck,
            "recall_check_structured_data": rcheck2,
        }
        if sweep:
            line["L_pq_sweep"] = sweep
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
