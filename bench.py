#!/usr/bin/env python3
"""bench.py -- QPS @ recall@10 of the RoarGraph search hot path on MI355X (BASELINE.json metric #1), and the
ground-truth build rate (metric #2).

Workload (BASELINE.json configs[1], "t2i-10M d=200 IP, top-10, L_pq sweep 10-2000, 1xMI355X search kernel"):
a 10M x 200 inner-product base and a GENUINE RoarGraph index over it, both made inside the run because no dataset or
prebuilt index can be downloaded here: structured synthetic embeddings (low intrinsic dimension, out-of-distribution
queries -- roargraph_amd/synth.py), ground truth of 2M training queries by K2 (sharded over the ranks when N > 1),
graph construction with the reference's parameters (M_sq=100, M_pjbp=35, L_pjpq=500; phase 3 on the GPU), then

  * an L_pq sweep over the reference's evaluation list (README.md:110-120): QPS, recall@10, evaluations, hops, GB/s,
    % of the 8.0 TB/s peak and of the 6.29 TB/s measured stream ceiling, x the CPU baseline;
  * the headline: `value` = queries/s at the SMALLEST L_pq of the sweep whose recall@10 >= 0.90, timed over exactly
    --steps batches of 10,000 queries between barriers, rg_search_wait included;
  * roofline of that launch (HIP events on the launch stream) and the CPU baseline ON THE SAME INDEX AND QUERIES
    (oracle/_ref/rg_ref = the reference's own distance/queue/visited code, 16 threads), whose ids must equal the GPU's;
  * `roofline_worstcase`: the same base under a random out-degree-40 graph at L_pq = 500 (no locality at all: every
    neighbour is fresh, the gather is pure random 800-byte reads) -- the round-1 headline, kept as the stress case;
  * `cpu_baseline_config1`: BASELINE configs[0], a 100K-row subset with its own index, L_pq = 50, ONE CPU thread;
  * `gt_build`: K2 over 65,536 queries x the 10M base (sharded over the ranks + all-to-all + K3), distances/s.

  python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1 without a launcher: re-executes itself under
                                                              torch.distributed.run, one rank per GPU, 127.0.0.1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Every step searches a DIFFERENT seeded batch of 10,000 queries (steps + warm-ups distinct batches, rotated): no launch
replays the rows the previous one left in the L2 / Infinity Cache.  Hub rows near the entry point are shared by every
query of every batch -- that reuse is the workload's own -- and `roofline.distinct_rows_frac` says how much of a
launch's evaluations are first touches of a row (the share that HBM itself must serve).

Multi-GPU: the index is replicated, every rank searches its own batch of queries (independent units, no data-path
collective); value = total queries / max-over-ranks time; scaling = weak.  The index is built once (training-query
ground truth sharded over all ranks, construction on rank 0) and broadcast.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import faulthandler  # noqa: E402
try:
    faulthandler.enable()  # a run that dies (SIGABRT of a GPU fault, SIGSEGV) leaves the Python stack of every thread on stderr
except Exception:  # noqa: BLE001  (a stderr without a file descriptor: imported under a test's capture)
    pass

# the reference's evaluation list (README.md:118) thinned to the points that shape the curve
SWEEP_DEFAULT = "10,20,30,40,50,60,80,100,150,200,300,500,700,1000,1500,2000"
# ... and the list itself, all 56 points (`--sweep readme`)
SWEEP_README = ("10,15,20,25,30,35,40,45,50,55,60,65,70,75,80,85,90,95,100,105,110,115,120,125,130,135,140,145,150,155,160,165,170,175,180,185,"
                "190,195,200,220,240,260,280,300,350,400,450,500,550,600,700,800,900,1000,1500,2000")


from benchlib import legs  # noqa: E402
from benchlib.report import compact_line  # noqa: E402
from benchlib.workloads import MALL_ROWS, _mem_available_gb, pmc_traffic, progress  # noqa: E402


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
    ap.add_argument("--metric", default="ip")
    ap.add_argument("--data", default="lowrank", help="lowrank (default: structured embeddings, where a graph index reaches high "
                    "recall) | mixture (1,000 clusters in the latent space: low reuse between queries) | gaussian (no structure at all)")
    ap.add_argument("--rank", type=int, default=32, help="latent rank of --data lowrank")
    ap.add_argument("--graph", default="roargraph", help="roargraph (default: genuine index built in the run) | random (out-degree --deg)")
    ap.add_argument("--real-index", action="store_true", help="(kept for older command lines) same as --graph roargraph")
    ap.add_argument("--deg", type=int, default=40)
    ap.add_argument("--train", type=int, default=0, help="training queries of the index build (default nb/5)")
    ap.add_argument("--L", type=int, default=0, help="beam width of the timed headline; 0 = the smallest L_pq of the sweep with recall@10 >= --target-recall")
    ap.add_argument("--target-recall", type=float, default=0.90)
    ap.add_argument("--sweep", default=SWEEP_DEFAULT, help="comma list of L_pq values (empty = none); `readme` = the reference's own 56-point "
                    "evaluation list (README.md:118)")
    ap.add_argument("--configs", default="rank128,webvid,laion", help="comma list of the side blocks of the default run, each a smaller build + search "
                    "of its own with roofline and cpu_baseline (rank 0, N = 1): rank128 = the headline's shape on harder data (latent rank 128: 0.9 recall "
                    "needs L_pq 300), mixture = the headline's shape on data with LOW REUSE between queries (10,000 clusters in a rank-128 latent space; "
                    "easy for the search: not in the default run), webvid = BASELINE configs[4] end to end (2.5M x 512 IP: ground truth -> build -> "
                    "search), laion = BASELINE configs[3] shape (d = 512 L2 top-100) at the size --laion-nb; empty = none")
    ap.add_argument("--side-nb", type=int, default=0, help="rows of EVERY side block (tests: small sets); 0 = their own sizes")
    ap.add_argument("--rank128-nb", type=int, default=10_000_000, help="rows of the rank128 / mixture side blocks (default: the headline's size)")
    ap.add_argument("--laion-nb", type=int, default=2_000_000, help="rows of the laion-shaped side block (the full 10M x 512 run: "
                    "python bench.py --nb 10000000 --dim 512 --metric l2 --k 100 --configs '')")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of each CPU baseline sample (0 = skip)")
    ap.add_argument("--visited", type=int, default=2,
                    help="2 = library default (LDS visited filter + id log + exact distinct count, adaptive to the exact HBM words; "
                         "every output bit-exact); 1 = LDS filter only (cmps = evaluations performed); 0 = visited words in HBM")
    ap.add_argument("--set", default="", help="comma list of knob=value passed to rg_index_set (tuning experiments)")
    ap.add_argument("--row-stride", type=int, default=0, help="experiment: base rows padded to this many floats (256 with RG_SPLIT_ROWS=0: eight whole "
                    "128-B lines per d = 200 row instead of the split copy); 0 = dim")
    ap.add_argument("--no-worstcase", action="store_true", help="skip the random-graph block")
    ap.add_argument("--no-fast", action="store_true", help="skip the opt-in non-parity modes")
    ap.add_argument("--no-two-streams", action="store_true", help="skip the two-stream block (profiling: its overlapped launches would "
                    "blur the per-launch kernel durations of the trace)")
    ap.add_argument("--gt-nq", type=int, default=65536, help="queries of the ground-truth (K2) leg; 0 = skip")
    ap.add_argument("--gt-K", type=int, default=100)
    ap.add_argument("--k2-d512-nb", type=int, default=3_000_000, help="base rows of the d = 512 K2 launches of the ground-truth leg (ip and l2, 65,536 and "
                    "10,000 queries each: the truth of BASELINE configs[3] / [4]); 0 = skip")
    ap.add_argument("--config1-nb", type=int, default=100_000, help="rows of the BASELINE configs[0] subset (0 = skip)")
    ap.add_argument("--data-root", default="", help="directory with the reference's files (README.md:93-117): base.10M.fbin, "
                    "query.10k.fbin and, optionally, t2i_10M_roar.index (else query.train.10M.fbin to build one from). Overrides the "
                    "synthetic set; --nb / --dim / --nq are taken from the files. Names: --base-file / --query-file / --train-file / --index-file")
    ap.add_argument("--base-file", default="base.10M.fbin")
    ap.add_argument("--query-file", default="query.10k.fbin")
    ap.add_argument("--train-file", default="query.train.10M.fbin")
    ap.add_argument("--index-file", default="t2i_10M_roar.index")
    ap.add_argument("--no-retry", "--in-process", dest="no_retry", action="store_true", help="one-GPU runs: do the work in THIS process (default: in a "
                    "child process whose death ends the bench with its status and a post-mortem of the GPU fault; there is no second attempt)")
    ap.add_argument("--full-out", default="", help="file the FULL record is written to (default: bench_full.json beside this script, plus "
                    "gpurun_out/bench_full.json when that directory exists); stdout carries one compact line")
    ap.add_argument("--index-cache", default="", help="file the built graph is kept in (profiling: the rocprofv3 passes of one box "
                    "re-use the index the first pass built; the data set is seeded, so it is the same base)")
    return ap.parse_args()


def supervise():
    """One-GPU runs are carried out by a CHILD process of this script (benchlib/supervise.py); this GPU-less parent passes its one
    JSON line on.  No second attempt (round 5 had one): a child that dies ends the bench with its status, and the parent prints the
    post-mortem of a GPU fault -- which buffer the address belonged to -- from the report the library wrote.  --no-retry (kept for older
    command lines) = --in-process: the work happens in this process."""
    from benchlib import supervise as sv
    sv.supervise(os.path.abspath(__file__), sys.argv[1:], ROOT)


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: N ranks of this script under torch.distributed.run, one per
    GPU, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    import socket
    import subprocess
    from roargraph_amd._lib import lib
    ndev = lib().rg_device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.backend == "nccl" and ndev < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d device(s) visible (RCCL needs one GPU per rank; --backend gloo runs "
                         "several ranks on one GPU to exercise the control flow only)" % (args.gpus, ndev))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.sweep == "readme":
        args.sweep = SWEEP_README
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and "RG_BENCH_CHILD" not in os.environ and not args.no_retry:
        supervise()
    if "RG_BENCH_CHILD" in os.environ:      # the child of supervise(): it must not outlive a parent that was killed outright
        try:
            import ctypes
            import signal
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)      # PR_SET_PDEATHSIG
        except Exception:  # noqa: BLE001
            pass
    import torch
    import torch.distributed as dist
    from roargraph_amd import build, groundtruth, synth
    from roargraph_amd._lib import lib
    from roargraph_amd.index import IndexBipartite

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if lib().rg_device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch as many ranks as --gpus says" % (args.gpus, world))
    if args.backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d: only %d device(s) visible" % (world, torch.cuda.device_count()))
    local = local % torch.cuda.device_count()   # (control-flow tests run several ranks on one GPU with --backend gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where collectives on small tensors live
    stream = torch.cuda.current_stream().cuda_stream
    roar = args.graph == "roargraph" or args.real_index
    t_all = time.perf_counter()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # everything the legs share (benchlib/legs.py: each leg reads its inputs from C and leaves its results there)
    from types import SimpleNamespace
    C = SimpleNamespace(args=args, torch=torch, dist=dist, dev=dev, cdev=cdev, stream=stream, rank=rank, world=world, local=local, roar=roar,
                        sync_all=sync_all, synth=synth, groundtruth=groundtruth, build=build, lib=lib, IndexBipartite=IndexBipartite, t_all=t_all)
    legs.make_data_and_graph(C)
    legs.open_index_and_batches(C)
    legs.sweep_leg(C)
    S, L_star, sweep = C.S, C.L_star, C.sweep

    # ---- the timed headline: exactly --steps batches at L_star between barriers, the wait included; every warm-up and
    # every step searches another batch (rotation over `nbatch` distinct ones)
    progress("headline")
    for _ in range(args.warmup):
        S.run(L_star)
    S.wait()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    used = []
    e_beg, e_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync_all()
    t0 = time.perf_counter()
    e_beg.record()
    for a, b in evs:
        a.record()
        used.append(S.run(L_star))
        b.record()
    S.wait()
    e_end.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    # per launch = the HIP-event span of the timed region (first enqueue ... rg_search_wait returned) over its steps; the
    # per-enqueue pairs are reported beside it
    k1_on_stream_ms = [a.elapsed_time(b) for a, b in evs]
    kernel_ms = [e_beg.elapsed_time(e_end) / args.steps]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    qps = args.nq * args.steps * world / elapsed
    head = S.point(L_star, float(np.mean(kernel_ms)), used)
    C.qps, C.head, C.elapsed = qps, head, elapsed
    legs.replay_and_checks(C)
    replay_ms, replay_alg, reuse, forms_head = C.replay_ms, C.replay_alg, C.reuse, C.forms_head
    kavg = float(np.mean(kernel_ms)) / 1e3
    alg_bytes = head["mean_evals"] * args.nq * 4.0 * args.dim
    achieved = alg_bytes / kavg / 1e9
    wl_key = {"nb": args.nb, "dim": args.dim, "nq": args.nq, "k": args.k, "metric": args.metric, "data": args.data, "rank": args.rank,
              "graph": "roargraph" if roar else "random", "L": L_star, "visited": args.visited}
    del S      # (the side blocks close the main index and hand its memory back: nothing here may keep it alive)
    for leg in (legs.host_form_leg, legs.two_streams_leg, legs.opt_in_modes_leg, legs.cpu_baselines_leg, legs.worst_case_leg,
                legs.ground_truth_leg, legs.side_blocks_leg):
        leg(C)
    host_form, two_streams, fast, shared = C.host_form, C.two_streams, C.fast, C.shared
    cpu, cpu1, cpu_cfg1, gt_check, worst, gt = C.cpu, C.cpu1, C.cpu_cfg1, C.gt_check, C.worst, C.gt
    side_blocks, mem_stats_main, n_query_batches = C.side_blocks, C.mem_stats_main, C.n_query_batches
    data_desc, graph_desc, t_gt, t_build = C.data_desc, C.graph_desc, C.t_gt, C.t_build
    traffic, traffic_src = pmc_traffic(wl_key) if rank == 0 else (None, None)
    shape_name = {(10_000_000, 200, "ip"): "t2i-10M-shaped", (10_000_000, 512, "l2"): "laion-10M-shaped",
                  (2_500_000, 512, "ip"): "webvid-2.5M-shaped"}.get((args.nb, args.dim, args.metric), "%dx%d" % (args.nb, args.dim))
    if rank == 0:
        if cpu and cpu.get("value"):
            # x CPU for the sweep: the CPU baseline is measured at the headline L_pq only (bounded run time); its cost per
            # evaluation carries over, so other points are scaled by their evaluation counts
            per_eval = 1.0 / (cpu["value"] * cpu["mean_evals"])
            for p in sweep:
                p["x_cpu_16_threads_est"] = p["qps"] * p["mean_evals"] * per_eval
            cpu["gpu_over_cpu"] = qps / cpu["value"]
        line = {
            "metric": "QPS @ recall@10 >= %.2f, %s d=%d %s (search, top-%d, smallest L_pq reaching it: %d)"
                      % (args.target_recall, shape_name, args.dim, args.metric.upper(), args.k, L_star),
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "files" if args.data_root else "synthetic",
            "config": {"workload": "%s: base %dx%d fp32 %s, %d queries/GPU/step (a different seeded batch every step), top-%d, L_pq=%d, %s, %s (replicated per GPU)"
                                   % (shape_name, args.nb, args.dim, args.metric, args.nq, args.k, L_star, data_desc, graph_desc),
                       "parallelism": "query-sharded x%d, index replicated" % world,
                       "distinct_query_batches": n_query_batches,
                       "L_pq": L_star, "recall_at_10": head["recall_at_10"], "target_recall": args.target_recall,
                       "visited": {2: "default: lds-filter + id log + exact distinct count, adaptive to the exact HBM words where a timed "
                                      "trial finds them faster (ids/dists/hops/cmps bit-exact vs the HBM-visited mode, checked in this run)",
                                   1: "lds-filter only (ids/dists/hops bit-exact; cmps = evaluations performed)",
                                   0: "exact visited words in HBM"}[args.visited],
                       "mean_evals_per_query": head["mean_evals"], "mean_hops": head["mean_hops"],
                       "setup_seconds": {"train_gt": t_gt, "build": t_build, "total_run": time.perf_counter() - t_all}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src, "traffic_key": wl_key,
                         # the two bounds of "how much of frac did HBM itself deliver" at the top level: frac_hbm_only = the
                         # same base under a random graph (no row is read twice: every byte comes from HBM; L_pq = 500), the one
                         # driver-timed point where the memory system's figure and HBM's coincide; frac_cache_served = the share of the
                         # headline launch's row reads that go to rows an ideal 256-MiB cache could hold (measured from its id logs)
                         "frac_hbm_only": worst["frac"] if worst else None,
                         "frac_cache_served": reuse.get("share_of_reads_to_top_%d_rows" % MALL_ROWS) if reuse else None,
                         "kernel": "rg_search_kernel (exact visited set in LDS: no second kernel)" if forms_head.get("batches_lset") else
                                   "rg_search_kernel (+ rg_distinct_kernel in visited mode 2)", "kernel_ms_avg": kavg * 1e3,
                         "kernel_forms_of_the_batches_so_far": forms_head,
                         "kernel_ms_avg_is": "HIP-event span of the timed region (first enqueue on the launch stream ... rg_search_wait "
                                             "returned) / steps",
                         "k1_ms_per_enqueue_on_launch_stream": float(np.mean(k1_on_stream_ms)),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "frac_of_measured_stream_ceiling_6290": achieved / 6290.0,
                         # how much of `achieved` HBM itself had to serve.  distinct_rows_frac: the share of a launch's row reads
                         # that are FIRST touches of a row within the launch (rg_search_reuse_stats over the id logs); the rest
                         # re-reads rows other queries of the same launch read moments earlier (the neighbourhood of the entry
                         # point), which the Infinity Cache can serve -- FETCH_SIZE counts those reads, no counter separates them.
                         "distinct_rows_frac": reuse.get("distinct_rows_frac") if reuse else None,
                         # cache_served_frac_ceiling: the share of the launch's reads that go to its 349,525 most read rows --
                         # what a 256-MiB cache can hold; no cache of that size could have served more.  hbm_frac_floor =
                         # frac x (1 - that): the part of the algorithmic rate HBM itself certainly delivered.  The counters
                         # of the same command (L2 hit rate 6.6 %, FETCH_SIZE calibrated x2.000) are in profiles/r03/.
                         "cache_served_frac_ceiling": reuse.get("share_of_reads_to_top_%d_rows" % MALL_ROWS) if reuse else None,
                         "hbm_frac_floor": (achieved / 8000.0 * (1.0 - reuse["share_of_reads_to_top_%d_rows" % MALL_ROWS]))
                         if reuse and ("share_of_reads_to_top_%d_rows" % MALL_ROWS) in reuse else None,
                         "reuse": reuse,
                         "replay_same_batch": {"what": "the same launch replaying ONE batch back to back (round 2's protocol): the rows of "
                                                       "the previous launch are still in the Infinity Cache; not `value`",
                                               "kernel_ms_avg": replay_ms, "frac": replay_alg / (replay_ms / 1e3) / 1e9 / 8000.0
                                               if replay_ms > 0 else None}},
            "cpu_baseline": cpu,
            "cpu_baseline_1_thread": cpu1,
            "cpu_baseline_config1": cpu_cfg1,
            "recall_truth_crosscheck": gt_check,
            "L_pq_500": next((p for p in sweep if p["L_pq"] == 500), None),
            "L_pq_sweep": sweep,
            "roofline_worstcase": worst,
            "host_form_pcie_inclusive": host_form,
            "two_streams_pipelined": two_streams,
            "non_parity_modes": fast,
            "exact_opt_in_modes": shared,
            "gt_build": gt,
            "configs": side_blocks,
            "device_memory": mem_stats_main,
            "host_memory_GB": {"MemAvailable_at_end": round(_mem_available_gb(), 1)},
        }
        # The full record goes to a FILE (--full-out; default bench_full.json beside this script, and a copy under gpurun_out/
        # when that directory exists); stdout carries exactly ONE compact JSON line (a few KB) with the contract's keys, the
        # roofline and cpu_baseline objects and a summary row per sweep point / side block.  (Round 4 printed the full record
        # as the line: 25 KB, which the driver could not parse.)
        full_path = args.full_out or os.path.join(ROOT, "bench_full.json")
        wrote = []
        for pth in [full_path] + ([os.path.join(ROOT, "gpurun_out", "bench_full.json")] if not args.full_out and os.path.isdir(os.path.join(ROOT, "gpurun_out")) else []):
            try:
                with open(pth, "w") as fh:
                    json.dump(line, fh)
                wrote.append(os.path.relpath(pth, ROOT) if pth.startswith(ROOT) else pth)
            except OSError as e:
                print("[bench] could not write %s: %r" % (pth, e), file=sys.stderr)
        out = json.dumps(compact_line(line, wrote), separators=(",", ":"))
        assert len(out) < 8192, "the final line must stay small enough for any consumer (%d bytes)" % len(out)
        print(out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

