"""What bench.py measures with, beside the contract line: the searcher with its distinct query batches and HIP-event timing, the CPU
baselines (oracle/_ref/rg_ref = the reference's own headers, or the C port), the committed PMC traffic of a workload, row-reuse
statistics, and the side blocks (smaller workloads built and searched end to end inside the default run).  Moved out of bench.py in
round 6: bench.py keeps the contract -- flags, the timed region, the line."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_search_baseline(base_np, off, nbrs, ep, q_np, ids_gpu, metric, k, L, threads_list, budget_s):
    """The CPU path on a bounded sample of the same workload, once per entry of threads_list, each checked against the
    GPU's ids (an AssertionError here is a parity failure and ends the run).  oracle/_ref/rg_ref when it can run (the
    reference's own headers), else the AVX-512 restatement oracle/librg_oracle.so."""
    from oracle import pyoracle as po
    from roargraph_amd import io
    po.build() if not os.path.exists(po.LIB_PATH) else None
    nq = q_np.shape[0]
    po.use_avx512(True)
    use_ref = po.have_ref() and _mem_available_gb() > 3.0 * base_np.nbytes / 1e9
    outs = []
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        bf, qf, gf = (os.path.join(td, x) for x in ("b.fbin", "q.fbin", "g.index"))
        if use_ref:
            io.write_fbin(bf, base_np)
            io.write_index(gf, off, nbrs, ep)
        for threads in threads_list:
            out = {"unit": "QPS", "cores": threads, "host_cores": os.cpu_count() or 1, "L_pq": L}
            pilot = min(nq, 2 * threads)     # sized with the C port
            t0 = time.time()
            r = po.search(base_np, metric, off, nbrs, ep, q_np[:pilot], k, L, nthreads=threads)
            dt = max(time.time() - t0, 1e-6)
            assert (r[0] == ids_gpu[:pilot]).all(), "CPU oracle and GPU disagree on the bench workload (L_pq=%d)" % L
            n = int(min(nq, max(pilot, budget_s * pilot / dt)))
            n = max(threads, n - n % threads)
            if use_ref:
                io.write_fbin(qf, q_np[:n])
                # the reference's loop issues two software prefetches per neighbour (index_bipartite.cpp:2374-2375) and one
                # prefetch_vector of the entry point (:2324): that form is `value`; the loop without them (what round 2
                # timed) is recorded beside it
                ids, _, cmps, _, qps = po.ref_search(bf, gf, qf, metric, k, L, threads=threads, prefetch=True)
                assert (ids == ids_gpu[:n]).all(), "reference-header driver and GPU disagree on the bench workload (L_pq=%d)" % L
                ids_np, _, _, _, qps_np = po.ref_search(bf, gf, qf, metric, k, L, threads=threads, prefetch=False)
                assert (ids_np == ids).all()
                out.update(value=qps, value_without_prefetch=qps_np, kind="reference", mean_evals=float(np.mean(cmps)),
                           sample="%d queries, %d OpenMP thread(s), oracle/_ref/rg_ref (reference distance.h/neighbor.h/"
                                  "visited_list_pool.h; search loop restated with the reference's software prefetches, "
                                  "index_bipartite.cpp:2324,2374-2375), ids equal the GPU's" % (n, threads))
            else:
                t0 = time.time()
                r = po.search(base_np, metric, off, nbrs, ep, q_np[:n], k, L, nthreads=threads)
                dt = time.time() - t0
                assert (r[0] == ids_gpu[:n]).all(), "CPU oracle and GPU disagree on the bench workload (L_pq=%d)" % L
                out.update(value=n / dt, kind="port", mean_evals=float(np.mean(r[2])),
                           sample="%d queries, %d OpenMP thread(s), oracle/librg_oracle.so (avx512=%s), ids equal the GPU's"
                                  % (n, threads, bool(po.have_avx512())))
            outs.append(out)
    return outs


def gt_cpu_baseline(base, gq, args):
    """CPU baseline of the ground-truth leg: oracle/gt_numpy.py (blocked SGEMM on all host cores + per-query top-K, the
    shape of the reference's compute_groundtruth) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gt_numpy
    nbs, nqs = min(args.nb, 1_000_000), min(int(gq.shape[0]), 2048)
    hb = base[:nbs].cpu().numpy()
    hq = gq[:nqs].cpu().numpy()
    gt_numpy.groundtruth_blocked(hb[:65536], hq[:64], args.metric, args.gt_K)   # warm the BLAS threads
    t0 = time.perf_counter()
    gt_numpy.groundtruth_blocked(hb, hq, args.metric, args.gt_K)
    dt = time.perf_counter() - t0
    return {"value": float(nbs) * float(nqs) / dt, "unit": "distances/s", "cores": os.cpu_count() or 1, "kind": "port",
            "sample": "%d queries x %d base rows, K=%d, numpy/OpenBLAS SGEMM + argpartition per 131072-row block "
                      "(oracle/gt_numpy.py), %.1f s" % (nqs, nbs, args.gt_K, dt)}


def pmc_traffic(key):
    """HBM bytes per launch of the search kernel from the committed rocprofv3 PMC passes (profiles/*/search_traffic*.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE with the gfx950 x2 correction).  Counters cannot be read from
    inside the timed process, so the figure is reported only when a committed profile is of the workload being benched."""
    import glob
    paths = glob.glob(os.path.join(ROOT, "profiles", "**", "search_traffic*.json"), recursive=True)
    # newest round first (profiles/r04/... before profiles/r03/final/...), inside a round the deepest ("final") first
    for path in sorted(paths, key=lambda q: (os.path.relpath(q, ROOT).split(os.sep)[1], q.count(os.sep), q), reverse=True):
        try:
            t = json.load(open(path))
        except Exception:
            continue
        for ent in (t if isinstance(t, list) else [t]):
            if ent.get("workload") == key and ent.get("fetch_bytes_corrected"):
                total = float(ent["fetch_bytes_corrected"]) + float(ent.get("write_bytes") or 0.0)
                return total, "%s (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction %.4g B + WRITE_SIZE %.4g B per launch)" % (
                    os.path.relpath(path, ROOT), ent["fetch_bytes_corrected"], ent.get("write_bytes") or 0.0)
    return None, None


class Searcher:
    """One index + B distinct query batches, each with its own result buffers; timing helpers.  run() without a batch
    number takes the next one in rotation, so consecutive launches never search the same queries."""

    def __init__(self, torch, index, qs, k, dim, stream, gts):
        self.t, self.ix, self.qs, self.k, self.dim, self.stream = torch, index, list(qs), k, dim, stream
        self.gts = list(gts) if gts is not None else [None] * len(self.qs)
        dev, nq = self.qs[0].device, self.qs[0].shape[0]
        self.nq = nq
        self.out = [dict(ids=torch.zeros((nq, k), dtype=torch.int32, device=dev), dists=torch.zeros((nq, k), dtype=torch.float32, device=dev),
                         cmps=torch.zeros(nq, dtype=torch.int32, device=dev), hops=torch.zeros(nq, dtype=torch.int32, device=dev)) for _ in self.qs]
        self.cursor = 0
        self.depth_settled = {}

    def run(self, L, b=None):
        if b is None:
            b = self.cursor
            self.cursor = (self.cursor + 1) % len(self.qs)
        o = self.out[b]
        self.ix.search_dev(self.qs[b], self.k, L, o["ids"], o["dists"], o["cmps"], o["hops"], stream=self.stream)
        return b

    def wait(self):
        self.ix.search_wait(self.stream)

    def timed(self, L, reps=3, settle=3):
        """(average milliseconds per batch, batches timed) over `reps` launches (HIP events on the launch stream), after
        `settle` untimed batches (that is where the adaptive default decides between its two exact forms)."""
        t = self.t
        for _ in range(settle):
            self.run(L); self.wait()
        if L not in self.depth_settled or self.depth_settled[L] < reps:
            # the timed launches below are enqueued back to back: the first time `reps` batches are in flight on the stream the
            # library allocates the per-batch state of the 2nd, 3rd ... (a hipMalloc between the event records of that batch:
            # 2.4 - 6.7 ms once) -- let that happen here
            for _ in range(reps):
                self.run(L)
            self.wait()
            self.depth_settled[L] = reps
        ev = [(t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)) for _ in range(reps)]
        used = []
        names = ("batches_lset", "batches_filter_log", "batches_exact_hbm", "batches_filter_only")
        before = [self.ix.stat(n_) for n_ in names]
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        e0.record()
        for a, b in ev:
            a.record(); used.append(self.run(L)); b.record()
        self.wait()
        # the figure is the whole region -- first enqueue to the end of rg_search_wait (whatever the library ran for these
        # batches, on the launch stream or beside it, is done when the closing event is recorded) -- over the batches in it
        e1.record(); e1.synchronize()
        self.last_reps_ms = [a.elapsed_time(b) for a, b in ev]          # per enqueue, on the launch stream (K1 and what it waited for)
        # which kernel form the timed launches ran in (counters of the library) and the hub bitmap of the last one
        self.last_forms = {n_[8:]: self.ix.stat(n_) - b0 for n_, b0 in zip(names, before) if self.ix.stat(n_) - b0}
        self.last_forms["hub_bits_log2"] = self.ix.stat("hub_m_last")
        return e0.elapsed_time(e1) / reps, used

    def point(self, L, ms, used):
        """One row of the report: `used` = the batches the timing ran (their buffers hold the results at this L)."""
        from roargraph_amd import index as ixmod
        used = sorted(set(used))
        mc = float(np.mean([self.out[b]["cmps"].float().mean().item() for b in used]))
        mh = float(np.mean([self.out[b]["hops"].float().mean().item() for b in used]))
        rec = rec_k = None
        if self.k >= 10 and all(self.gts[b] is not None for b in used):
            rec = float(np.mean([ixmod.recall(self.out[b]["ids"].cpu().numpy().view(np.uint32), self.gts[b], 10) for b in used]))
            rk = getattr(self, "recall_k", 10)      # the reference's recall@k over all k results (test_search_roargraph.cpp:23-36)
            rec_k = rec if rk == 10 else float(np.mean([ixmod.recall(self.out[b]["ids"].cpu().numpy().view(np.uint32), self.gts[b], rk) for b in used]))
        gbps = self.nq * mc * 4 * self.dim / (ms / 1e3) / 1e9
        return {"L_pq": L, "qps": self.nq / (ms / 1e3), "ms_per_batch": ms, "recall_at_10": rec, "recall_at_k": rec_k, "mean_evals": mc, "mean_hops": mh,
                "distinct_batches": len(used), "GBps": gbps, "pct_of_8000": 100.0 * gbps / 8000.0, "pct_of_6290": 100.0 * gbps / 6290.0}


MALL_ROWS = 349525      # rows of 768 B the 256-MiB Infinity Cache can hold


def reuse_of_last_launch(torch, index, stream, nb, nq, dev, full=False):
    """First touches and popularity of the rows the last default-mode launch on `stream` read (rg_search_reuse_stats over its
    id logs); None when that launch ran on the exact words (no logs)."""
    try:
        counts = torch.zeros(nb, dtype=torch.int32, device=dev)
        ev_n, dr_n = index.reuse_stats(stream, counts)
    except Exception:  # noqa: BLE001
        return None
    srt = torch.sort(counts, descending=True).values.double()
    cum = torch.cumsum(srt, 0) / max(float(ev_n), 1.0)
    out = {"distinct_rows_frac": dr_n / max(ev_n, 1), "share_of_reads_to_top_%d_rows" % MALL_ROWS: float(cum[min(MALL_ROWS, nb) - 1].item())}
    if full:
        out.update({"evaluations_performed": ev_n, "distinct_rows": dr_n,
                    # popularity: share of the launch's row reads that go to its H most read rows (H rows = H x 768 B)
                    "share_of_reads_to_top_rows": {str(h): float(cum[min(h, nb) - 1].item()) for h in (64, 1024, 16384, 131072, MALL_ROWS, 1048576)},
                    "rows_read_by_every_query": int((counts >= nq).sum().item())})
    return out



_T0 = time.perf_counter()


def progress(what):      # RG_BENCH_PROGRESS=1: stage marks on stderr (where a run that dies was)
    if os.environ.get("RG_BENCH_PROGRESS"):
        print("[bench %7.1f s] %s" % (time.perf_counter() - _T0, what), file=sys.stderr, flush=True)


def side_config(torch, dev, stream, name, nb, dim, metric, k, rank_latent, ntrain, nq, Ls, target, cpu_seconds, steps, what, frac_hbm_only=None, data="lowrank"):
    """One smaller workload end to end inside the default run: data -> ground truth of the training queries (K2) ->
    GPU-assisted RoarGraph construction -> a short L_pq sweep -> `steps` timed batches at the smallest L_pq reaching `target`
    recall@10 (recall@k for the top-100 shape) -> the reference loop on 16 host threads over the same index and queries (ids
    asserted equal).  Returns a block with its own `roofline` and `cpu_baseline`."""
    from roargraph_amd import build, groundtruth, synth
    from roargraph_amd.index import IndexBipartite
    progress("side block %s: start" % name)
    t_all = time.perf_counter()
    base, train, q, desc = synth.make_device_set(dev, 1234, nb, ntrain, nq, dim, data=data, rank=rank_latent, q_seed=99)
    t0 = time.perf_counter()
    ti, _ = groundtruth.groundtruth_distributed(base, 0, train, metric, 100)
    torch.cuda.synchronize()
    progress("side block %s: training truth done, building" % name)
    t_gt = time.perf_counter() - t0
    t0 = time.perf_counter()
    h_off, h_nbrs, ep = build.build_roargraph(synth.to_host(base), synth.to_host(ti).view(np.uint32), metric, 100, 35, 500,
                                              num_threads=int(os.environ.get("RG_BENCH_BUILD_THREADS", min(128, os.cpu_count() or 1))), device=dev.index or 0)
    t_build = time.perf_counter() - t0
    del train, ti
    off = synth.to_device(h_off.view(np.int64), dev)
    nbrs = synth.to_device(h_nbrs.view(np.int32), dev)
    progress("side block %s: built, opening" % name)
    torch.cuda.empty_cache()
    index = IndexBipartite.from_device(base, off, nbrs, ep, metric=metric)
    nbatch = 3
    qs = [q] + [synth.make_device_set(dev, 1234, 1024, 0, nq, dim, data=data, rank=rank_latent, q_seed=99 + 7919 * b)[2] for b in range(1, nbatch)]
    gts = []
    ti_q = torch.zeros((nq, 100), dtype=torch.int32, device=dev); tv_q = torch.zeros((nq, 100), device=dev)
    t0 = time.perf_counter()
    for qb in qs:
        groundtruth.gt_shard_dev(base, qb, metric, 100, 0, ti_q, tv_q, stream=stream); torch.cuda.synchronize()
        gts.append(ti_q.cpu().numpy().view(np.uint32).copy())
    t_gtq = time.perf_counter() - t0
    del ti_q, tv_q
    progress("side block %s: query truth done, sweep" % name)
    S = Searcher(torch, index, qs, k, dim, stream, gts)
    S.recall_k = k if k <= 100 else 10
    sweep = []
    for L in [x for x in Ls if x >= k]:
        ms, used = S.timed(L, reps=2, settle=2)
        sweep.append(S.point(L, ms, used))
    progress("side block %s: headline" % name)
    ok = [p["L_pq"] for p in sweep if (p["recall_at_k"] or 0.0) >= target]
    L_star = min(ok) if ok else max(p["L_pq"] for p in sweep)
    ms, used = S.timed(L_star, reps=steps, settle=1)
    head = S.point(L_star, ms, used)
    forms = {n_: index.stat(n_) for n_ in ("batches_lset", "batches_filter_log", "batches_exact_hbm")}
    head_forms = dict(S.last_forms)
    S.run(L_star, 0); S.wait()
    ids_head = S.out[0]["ids"].cpu().numpy().view(np.uint32).copy()
    # share of the headline launch's row reads that go to rows a 256-MiB cache can hold (one untimed launch in the logging form)
    reuse = None
    try:
        index.set("lset", 0); index.set("adaptive", 0)
        S.run(L_star, 0); S.wait()
        reuse = reuse_of_last_launch(torch, index, stream, nb, nq, dev)
        index.set("lset", -1); index.set("adaptive", 1)
    except Exception:  # noqa: BLE001
        reuse = None
    progress("side block %s: reuse statistics done" % name)
    cpu = None
    if cpu_seconds > 0:
        try:
            cpu = cpu_search_baseline(synth.to_host(base), h_off, h_nbrs, ep, qs[0].cpu().numpy(), ids_head, metric, k, L_star,
                                      [min(16, os.cpu_count() or 1)], cpu_seconds)[0]
            cpu["gpu_over_cpu"] = head["qps"] / cpu["value"] if cpu.get("value") else None
        except AssertionError:
            raise
        except Exception as e:  # noqa: BLE001
            cpu = {"value": None, "unit": "QPS", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    progress("side block %s: closing" % name)
    index.close()
    alg = head["mean_evals"] * nq * 4.0 * dim
    tr_, trs = pmc_traffic({"nb": nb, "dim": dim, "nq": nq, "k": k, "metric": metric, "data": data, "rank": rank_latent, "graph": "roargraph",
                            "L": L_star, "visited": 2})
    out = {"name": name, "what": what, "nb": nb, "dim": dim,
           "workload": "base %dx%d fp32 %s (%s), %d training queries, own RoarGraph index (M_sq=100 M_pjbp=35 L_pjpq=500, avg degree %.1f), %d queries/batch "
                       "(%d distinct batches), top-%d, L_pq=%d" % (nb, dim, metric, desc, ntrain, float(h_nbrs.size) / nb, nq, nbatch, k, L_star),
           "metric": "QPS @ recall@%d >= %.2f" % (S.recall_k, target), "value": head["qps"], "unit": "queries/s",
           "L_pq": L_star, "recall_at_k": head["recall_at_k"], "recall_k": S.recall_k, "mean_evals": head["mean_evals"], "mean_hops": head["mean_hops"],
           "seconds": {"train_ground_truth": t_gt, "construction": t_build, "query_ground_truth": t_gtq, "block_total": time.perf_counter() - t_all},
           "roofline": {"bound": "hbm", "achieved": head["GBps"], "peak": 8000.0, "unit": "GB/s", "frac": head["GBps"] / 8000.0,
                        "kernel_ms_avg": ms, "algorithmic_bytes_per_launch": alg, "traffic": tr_, "traffic_source": trs,
                        "frac_hbm_only": frac_hbm_only,
                        "frac_cache_served": reuse.get("share_of_reads_to_top_%d_rows" % MALL_ROWS) if reuse else None,
                        "distinct_rows_frac": reuse.get("distinct_rows_frac") if reuse else None,
                        "kernel_forms_of_the_timed_launches": head_forms,
                        "frac_of_measured_stream_ceiling_6290": head["GBps"] / 6290.0},
           "cpu_baseline": cpu, "kernel_forms_of_the_batches": forms,
           "L_pq_sweep": [{"L_pq": p["L_pq"], "qps": p["qps"], "recall_at_k": p["recall_at_k"], "mean_evals": p["mean_evals"], "pct_of_8000": p["pct_of_8000"]}
                          for p in sweep]}
    del S, index, base, off, nbrs, qs
    torch.cuda.empty_cache()
    return out
