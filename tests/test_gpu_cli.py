"""-m gpu: the C++ CLI twins (roargraph_amd/bin) against the oracle: same flags, same table/CSV columns as
tests/test_search_roargraph.cpp:190,231-236, same gt file layout as compute_groundtruth (README.md:70-74)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import small_set
from roargraph_amd import io

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "roargraph_amd", "bin")


@pytest.fixture(scope="module")
def bins():
    if not os.path.exists(os.path.join(BIN, "test_search_roargraph")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "roargraph_amd", "cli")])
    return BIN


def test_compute_groundtruth_then_search_cli(bins, oracle, tmp_path):
    metric, d, nb = "ip", 200, 3000
    base, q, off, nbrs, ep = small_set(metric, nb, d, nq=120)
    bf, qf, gf, gtf, csv = (str(tmp_path / x) for x in ("b.fbin", "q.fbin", "g.index", "gt.bin", "eval.csv"))
    io.write_fbin(bf, base); io.write_fbin(qf, q); io.write_index(gf, off, nbrs, ep)
    # ground truth through the CLI twin
    r = subprocess.run([os.path.join(bins, "compute_groundtruth"), "--data_type", "float", "--dist_fn", "mips",
                        "--base_file", bf, "--query_file", qf, "--gt_file", gtf, "--K", "100"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    gt_ids, gt_d = oracle.gt_load(gtf)                       # layout accepted by the reference's size rule
    ref_ids, _, ref_s = oracle.groundtruth_f64(base, q, metric, 100, nthreads=8)
    assert (gt_ids == ref_ids).mean() > 0.999
    assert np.allclose(gt_d, ref_s, rtol=1e-4, atol=1e-4)
    # search through the CLI twin, two L_pq values as a multitoken option
    r = subprocess.run([os.path.join(bins, "test_search_roargraph"), "--data_type", "float", "--dist", "ip",
                        "--base_data_path", bf, "--query_path", qf, "--gt_path", gtf,
                        "--projection_index_save_path", gf, "--L_pq", "20", "100", "--k", "10", "-T", "16",
                        "--evaluation_save_path", csv], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # stdout table: the reference's header line, character for character, and its row format (tests/golden/cli_table.json =
    # the stream statements of test_search_roargraph.cpp:190,231-232 evaluated by scripts/make_golden.py g5)
    fmt = json.load(open(os.path.join(ROOT, "tests", "golden", "cli_table.json")))
    lines = r.stdout.splitlines()
    assert fmt["header"].format(k=10) in lines, "header line differs from the reference's"
    row_re = re.compile("^" + re.sub(r"\\\{[^}]*\\\}", r"(?:[-+0-9.eE]+|nan|inf)", re.escape(fmt["row"]).replace("\\\t", "\t")) + "$")
    table = [l for l in lines if l.split() and l.split()[0] in ("20", "100")]
    assert len(table) == 2 and all(len(t.split("\t\t")) == 6 and float(t.split()[1]) > 0 for t in table), "six columns, as the reference"
    assert all(row_re.match(t) for t in table), (fmt["row"], table)
    rows = [l.split(",") for l in open(csv).read().strip().splitlines()]
    assert [int(x[0]) for x in rows] == [20, 100] and all(len(x) == len(fmt["csv_row"].split(",")) == 6 for x in rows)
    # --steady 1: the additive seventh column (a later pass, after the adaptive default has settled)
    r7 = subprocess.run([os.path.join(bins, "test_search_roargraph"), "--data_type", "float", "--dist", "ip",
                         "--base_data_path", bf, "--query_path", qf, "--gt_path", gtf,
                         "--projection_index_save_path", gf, "--L_pq", "20", "--k", "10", "--steady", "1"], capture_output=True, text=True)
    assert r7.returncode == 0, r7.stdout + r7.stderr
    assert fmt["header"].format(k=10) + "\tQPS_steady" in r7.stdout.splitlines()
    t7 = [l.split() for l in r7.stdout.splitlines() if l.split() and l.split()[0] == "20"]
    assert len(t7) == 1 and len(t7[0]) == 7 and float(t7[0][6]) > 0
    for row in rows:
        L = int(row[0])
        ids, _, cmps, hops = oracle.search(base, metric, off, nbrs, ep, q, 10, L, nthreads=4)
        assert float(row[2]) == pytest.approx(float(np.float32(cmps.astype(np.float32).sum() / 120)), rel=1e-5)
        assert float(row[5]) == pytest.approx(float(hops.mean()), rel=1e-5)
        assert float(row[4]) == pytest.approx(oracle.recall(ids, gt_ids, 10), abs=1e-6)
        assert float(row[1]) > 0
    # the same search with the index replicated on "two devices" (both device 0 here) and the queries sharded: the
    # recall / avg_visited / avg_hops columns do not change
    csv2 = str(tmp_path / "eval2.csv")
    r = subprocess.run([os.path.join(bins, "test_search_roargraph"), "--data_type", "float", "--dist", "ip",
                        "--base_data_path", bf, "--query_path", qf, "--gt_path", gtf,
                        "--projection_index_save_path", gf, "--L_pq", "20", "100", "--k", "10", "--devices", "0", "0",
                        "--evaluation_save_path", csv2], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rows2 = [l.split(",") for l in open(csv2).read().strip().splitlines()]
    assert [(x[0], x[2], x[4], x[5]) for x in rows2] == [(x[0], x[2], x[4], x[5]) for x in rows]


def test_cli_errors(bins, tmp_path):
    r = subprocess.run([os.path.join(bins, "test_search_roargraph"), "--data_type", "float"], capture_output=True, text=True)
    assert r.returncode != 0 and "is required but missing" in r.stderr
    base = np.zeros((10, 8), np.float32)
    bf = str(tmp_path / "b.fbin")
    io.write_fbin(bf, base)
    open(bf, "ab").write(b"\0" * 40)          # one extra row worth of bytes -> the reference's size check fires
    r = subprocess.run([os.path.join(bins, "compute_groundtruth"), "--data_type", "float", "--dist_fn", "l2",
                        "--base_file", bf, "--query_file", bf, "--gt_file", str(tmp_path / "o"), "--K", "2"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Data file size wrong!" in (r.stdout + r.stderr)


def test_end_to_end_pipeline_gt_build_search(oracle):
    """BASELINE config 5 in miniature: K2 ground truth of the training queries -> CPU graph build -> K1 search.
    The GPU search over the freshly built index must equal the oracle's search over the same index, and reach the
    recall the same pipeline reaches on the CPU."""
    from roargraph_amd import build, groundtruth, synth
    from roargraph_amd.index import IndexBipartite
    rng = np.random.default_rng(21)
    base = rng.standard_normal((6000, 200)).astype(np.float32)
    train = (rng.standard_normal((3000, 200)) * 0.5 + 0.3).astype(np.float32)
    q = (rng.standard_normal((200, 200)) * 0.5 + 0.3).astype(np.float32)
    knn, _ = groundtruth.compute_groundtruth(base, train, "ip", 100)
    ref_knn, _, _ = oracle.groundtruth_f64(base, train, "ip", 100, nthreads=16)
    assert (knn == ref_knn).mean() > 0.999
    off, nbrs, ep = build.build_roargraph(base, knn, "ip", 100, 35, 500, num_threads=1)
    ix = IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip")
    gt, _ = groundtruth.compute_groundtruth(base, q, "ip", 100)
    for L in (20, 200):
        got = ix.SearchRoarGraph(q, 10, L)
        want = oracle.search(base, "ip", off, nbrs, ep, q, 10, L, nthreads=8)
        assert all((a == b).all() for a, b in zip((got[0], got[2], got[3]), (want[0], want[2], want[3])))
        assert (got[1].view(np.uint32) == want[1].view(np.uint32)).all()
    assert oracle.recall(got[0], gt, 10) > 0.97
    ix.close()


def test_gpu_assisted_build(oracle, monkeypatch):
    """rg_build_roargraph_gpu: phase 3's beam searches run on the GPU (K1 in build mode).  With RG_BUILD_VERIFY every
    expansion list coming back from the GPU is compared, bit for bit, with the host search over the same frozen graph
    snapshot; the finished index must respect the degree bound and search as well as the all-CPU build."""
    from roargraph_amd import build
    monkeypatch.setenv("RG_BUILD_VERIFY", "1")
    rng = np.random.default_rng(33)
    for metric, d in (("ip", 200), ("l2", 64)):
        base = rng.standard_normal((5000, d)).astype(np.float32)
        train = (rng.standard_normal((2500, d)) * 0.5 + 0.3).astype(np.float32)
        q = (rng.standard_normal((150, d)) * 0.5 + 0.3).astype(np.float32)
        knn, _, _ = oracle.groundtruth_f64(base, train, metric, 100, nthreads=16)
        gt, _, _ = oracle.groundtruth_f64(base, q, metric, 100, nthreads=16)
        off_c, nbrs_c, ep_c = build.build_roargraph(base, knn, metric, 100, 24, 150, num_threads=8)
        off_g, nbrs_g, ep_g = build.build_roargraph(base, knn, metric, 100, 24, 150, num_threads=8, device=0, batch=700)
        assert ep_c == ep_g
        deg = np.diff(off_g.astype(np.int64))
        assert deg.max() <= 48 and nbrs_g.max() < 5000
        rc = oracle.recall(oracle.search(base, metric, off_c, nbrs_c, ep_c, q, 10, 100, nthreads=8)[0], gt, 10)
        rg_ = oracle.recall(oracle.search(base, metric, off_g, nbrs_g, ep_g, q, 10, 100, nthreads=8)[0], gt, 10)
        assert rg_ > rc - 0.03, (metric, rc, rg_)


@pytest.mark.parametrize("metric,d,nb,M,L", [("ip", 200, 1500, 16, 120), ("l2", 64, 1200, 12, 80)])
def test_gpu_assisted_build_one_node_at_a_time_equals_the_oracle_build(oracle, metric, d, nb, M, L):
    """rg_build_roargraph_gpu with batch = 1 and one host thread searches, prunes and links node after node, i.e. in the
    reference's one-thread order -- so the entry-point kernels, K1 in build mode and the occlusion-pruning kernel together
    must produce the index of the ORACLE's build (oracle/rg_oracle_build.c, written from the reference alone), byte for
    byte.  This is the oracle comparison of the f-1 / f-2 device pieces (the tests below compare them with the product's own
    host code)."""
    from roargraph_amd import build
    rng = np.random.default_rng(nb + d)
    r = 6
    A = (rng.standard_normal((r, d)) / np.sqrt(r)).astype(np.float32)
    base = (rng.standard_normal((nb, r)).astype(np.float32) @ A + 0.05 * rng.standard_normal((nb, d)).astype(np.float32))
    base[rng.integers(0, nb, 60)] = base[rng.integers(0, nb, 60)]            # exact duplicates: ties in distance
    train = ((0.3 + 0.5 * rng.standard_normal((600, r))).astype(np.float32) @ A).astype(np.float32)
    knn, _, _ = oracle.groundtruth_f64(base, train, metric, 60, nthreads=16)
    oracle.use_avx512(True)
    want = oracle.build_roargraph(base, knn, metric, 60, M, L)
    oracle.use_avx512(False)
    got = build.build_roargraph(base, knn, metric, 60, M, L, num_threads=1, device=0, batch=1)
    assert got[2] == want[2], "entry point"
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all(), "GPU-assisted build differs from the oracle's"


@pytest.mark.parametrize("metric,d,nb,M,L,batch", [("ip", 200, 7000, 16, 120, 0), ("l2", 64, 6000, 12, 80, 0), ("ip", 200, 5000, 24, 150, 700)])
def test_gpu_assisted_build_is_deterministic_and_equals_the_oracle_build(oracle, metric, d, nb, M, L, batch):
    """rg_build_roargraph_gpu for any number of host threads: the GPU computes the pruned lists of phase 1 and the searches +
    prunings of phase 3, the host replays the reverse edges list by list (phase1_replay, phase2_windows, link_batch).  The
    index must equal the ORACLE's build given the batch list of rg_build_schedule(nb, batch) byte for byte -- with 1, 3 and 8
    host threads -- and, for batch = 0, the all-host build at several threads."""
    from roargraph_amd import build
    rng = np.random.default_rng(nb + d)
    r = 6
    A = (rng.standard_normal((r, d)) / np.sqrt(r)).astype(np.float32)
    base = (rng.standard_normal((nb, r)).astype(np.float32) @ A + 0.05 * rng.standard_normal((nb, d)).astype(np.float32))
    base[rng.integers(0, nb, 100)] = base[rng.integers(0, nb, 100)]
    train = ((0.3 + 0.5 * rng.standard_normal((9000, r))).astype(np.float32) @ A).astype(np.float32)
    knn, _, _ = oracle.groundtruth_f64(base, train, metric, 60, nthreads=16)
    sched = build.build_schedule(nb, batch)
    assert int(sched.sum()) == nb
    oracle.use_avx512(True)
    want = oracle.build_roargraph(base, knn, metric, 60, M, L, sched=sched)
    oracle.use_avx512(False)
    for T in (1, 3, 8):
        got = build.build_roargraph(base, knn, metric, 60, M, L, num_threads=T, device=0, batch=batch)
        assert got[2] == want[2], "entry point"
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all(), f"GPU-assisted build, {T} host threads: differs from the oracle's"
    if batch == 0:
        host = build.build_roargraph(base, knn, metric, 60, M, L, num_threads=4)
        assert (host[0] == want[0]).all() and (host[1] == want[1]).all(), "all-host build at 4 threads differs"


@pytest.mark.parametrize("metric,d,M,L", [("ip", 200, 35, 500), ("l2", 512, 24, 150), ("ip", 104, 12, 100), ("l2", 200, 35, 300),
                                          ("ip", 200, 24, 600)])   # the last: lists longer than the kernel sorts -> host pruning
def test_gpu_pruning_equals_host_pruning(oracle, monkeypatch, metric, d, M, L):
    """The occlusion pruning of the phase-3 expansion lists (PruneProjectionBaseSearchCandidates, :1846-1940) on the GPU
    against Builder::prune_search on the host.  (i) RG_BUILD_VERIFY: every pruned list coming back from the GPU is compared
    with the host rule applied to the same expansion list (a difference fails the build).  (ii) With one host thread the
    whole GPU-assisted build is deterministic, so the index built with the GPU pruning must equal, edge for edge, the one
    built with RG_BUILD_HOST_PRUNE=1.  Structured data with repeated rows: ties in distance, heavy occlusion."""
    from roargraph_amd import build
    rng = np.random.default_rng(7 + d)
    nb, r = 6000, 8
    A = (rng.standard_normal((r, d)) / np.sqrt(r)).astype(np.float32)
    base = (rng.standard_normal((nb, r)).astype(np.float32) @ A + 0.05 * rng.standard_normal((nb, d)).astype(np.float32))
    base[rng.integers(0, nb, 300)] = base[rng.integers(0, nb, 300)]          # exact duplicates
    train = ((0.3 + 0.5 * rng.standard_normal((3000, r))).astype(np.float32) @ A).astype(np.float32)
    knn, _, _ = oracle.groundtruth_f64(base, train, metric, 100, nthreads=16)
    monkeypatch.setenv("RG_BUILD_VERIFY", "1")
    build.build_roargraph(base, knn, metric, 100, M, L, num_threads=8, device=0, batch=900)      # (i): raises on a mismatch
    monkeypatch.delenv("RG_BUILD_VERIFY")
    got = build.build_roargraph(base, knn, metric, 100, M, L, num_threads=1, device=0, batch=900)
    monkeypatch.setenv("RG_BUILD_HOST_PRUNE", "1")
    want = build.build_roargraph(base, knn, metric, 100, M, L, num_threads=1, device=0, batch=900)
    assert got[2] == want[2] and (got[0] == want[0]).all() and (got[1] == want[1]).all()
    assert np.diff(got[0].astype(np.int64)).max() <= 3 * M


def _bench_records(stdout, full_path):
    """(compact line, full record) of one bench.py run: stdout carries exactly one JSON line, under 8 KB, with the contract's
    keys (the driver could not parse the 25-KB line of round 4); the full record is the --full-out file."""
    import json
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    assert stdout.strip().splitlines()[-1] == lines[0], "the JSON line is the last line of stdout"
    assert len(lines[0]) < 8192, "compact line: %d bytes" % len(lines[0])
    c = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert "workload" in c["config"] and c["roofline"]["frac"] > 0 and c["roofline"]["bound"] == "hbm"
    with open(full_path) as fh:
        full = json.load(fh)
    assert full["value"] == pytest.approx(c["value"], rel=1e-4)
    return c, full


def test_bench_multi_rank_control_flow_on_one_gpu(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), here with two ranks on
    the one visible GPU and the gloo backend: rank/LOCAL_RANK handling, barriers, max-over-ranks timing, the sharded
    ground-truth leg with its all-to-all and K3 merge, one JSON line from rank 0."""
    import json
    import socket
    import sys
    with socket.socket() as sk:            # a port nobody is using right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--nb", "200000", "--nq", "512", "--gt-nq", "4096", "--cpu-seconds", "0", "--sweep", "20,100",
           "--no-worstcase", "--no-fast", "--config1-nb", "0", "--full-out", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    c, d = _bench_records(r.stdout, str(tmp_path / "full.json"))
    assert c["n_gpus"] == 2 and c["gt_build"]["value"] > 0 and [p[0] for p in c["sweep"]] == [20, 100, 500]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["gt_build"]["value"] > 0 and "sharded x2" in d["gt_build"]["metric"]
    assert "genuine RoarGraph index" in d["config"]["workload"] and "on 2 GPU(s)" in d["config"]["workload"]
    assert [p["L_pq"] for p in d["L_pq_sweep"]] == [20, 100, 500] and all(p["recall_at_10"] > 0.5 for p in d["L_pq_sweep"][1:])


def test_bench_gpus_flag_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 --backend gloo` with NO launcher around it: the script re-executes itself under
    torch.distributed.run with two ranks (here both on the one visible GPU) and reports the ranks that took part; with the
    RCCL backend it must refuse to start when fewer GPUs than ranks are visible."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
           "--nb", "200000", "--nq", "512", "--gt-nq", "4096", "--cpu-seconds", "0", "--sweep", "50", "--no-worstcase", "--no-fast",
           "--config1-nb", "0", "--full-out", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    _, d = _bench_records(r.stdout, str(tmp_path / "full.json"))
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert "query-sharded x2" in d["config"]["parallelism"]
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                           timeout=300, env=env)
        assert r.returncode != 0 and "device(s) visible" in (r.stdout + r.stderr)


def test_bench_one_gpu_small_run_reports_every_block(tmp_path):
    """The default one-GPU flow on a small set: distinct query batches per step, the replay figure, first-touch share of the
    row reads, both CPU loop forms, the native ground-truth leg (rg_groundtruth_rank, four streamed batches)."""
    import json
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--nb", "300000", "--nq", "1000", "--gt-nq", "65536",
           "--cpu-seconds", "1", "--sweep", "20,100", "--no-worstcase", "--no-fast", "--config1-nb", "0", "--configs", "webvid,laion", "--side-nb", "40000",
           "--k2-d512-nb", "200000",
           "--full-out", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    c, d = _bench_records(r.stdout, str(tmp_path / "full.json"))
    compact = c
    # the compact line: roofline.frac, cpu_baseline.value and one summary row per side block (what the driver records)
    assert c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["kind"] in ("reference", "port") and c["cpu_baseline"]["cores"] >= 1
    assert [x["name"] for x in c["configs_summary"]] == ["webvid", "laion"] and all(x["frac"] > 0 and x["qps"] > 0 for x in c["configs_summary"])
    assert c["device_memory"]["plain_fallbacks"] == 0 or os.environ.get("RG_BALANCED_ALLOC") == "0"
    # (round 6) beside the headline's cache-assisted fraction: the fraction at L_pq 500 of the same index, and the sentence that says what frac is
    assert c["roofline"]["frac_at_L500"] > 0 and "cache-assisted" in c["roofline"]["frac_is"] and c["bench_attempts"] == 1
    assert d["n_gpus"] == 1 and d["config"]["distinct_query_batches"] == 6
    # the side blocks (round 4): d = 512 IP end to end and d = 512 L2 top-100, each with its own roofline and cpu_baseline
    assert [c["name"] for c in d["configs"]] == ["webvid", "laion"]
    for c in d["configs"]:
        assert c["value"] > 0 and 0.0 < c["roofline"]["frac"] and c["recall_at_k"] > 0.5 and c["cpu_baseline"]["value"] > 0
        assert c["seconds"]["construction"] > 0 and c["seconds"]["train_ground_truth"] > 0
    assert d["configs"][1]["recall_k"] == 100 and "l2" in d["configs"][1]["workload"]
    assert d["roofline"]["frac_cache_served"] is not None and "frac_hbm_only" in d["roofline"]
    rf = d["roofline"]
    assert 0.0 < rf["frac"] and rf["distinct_rows_frac"] is not None and 0.0 < rf["distinct_rows_frac"] <= 1.0
    assert 0.0 < rf["cache_served_frac_ceiling"] <= 1.0 and rf["hbm_frac_floor"] is not None
    assert all("distinct_rows_frac" in p for p in d["L_pq_sweep"])
    assert rf["replay_same_batch"]["kernel_ms_avg"] > 0
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and (cb["kind"] != "reference" or cb["value_without_prefetch"] > 0)
    g = d["gt_build"]
    assert g["value"] > 0 and "4 query batches" in g["form"] and g["k2_device_resident"]["value"] > 0
    # (round 6) K2 at d = 512 in the record: both metrics, both batch sizes
    assert set(g["k2_d512"]) >= {"ip_65536", "ip_10000", "l2_65536", "l2_10000"} and all(g["k2_d512"][k] > 0 for k in ("ip_65536", "l2_10000"))
    assert compact["gt_build"]["k2_d512"]["l2_65536"] > 0


def test_bench_on_the_reference_file_layout(tmp_path):
    """bench.py --data-root: the reference's own file names (README.md:93-117).  First run: base + queries + training queries
    -> the index is built in the run; second run: an index file beside them is searched as it is.  Same base and queries, so
    both runs must report a sensible recall, and the second must say it used the file."""
    import json
    import sys
    from roargraph_amd import io, build, synth
    rng = np.random.default_rng(5)
    nb, d = 60000, 200
    A = (rng.standard_normal((16, d)) / 4.0).astype(np.float32)
    base = (rng.standard_normal((nb, 16)).astype(np.float32) @ A + 0.05 * rng.standard_normal((nb, d)).astype(np.float32))
    mk = lambda n: ((0.3 + 0.5 * rng.standard_normal((n, 16))).astype(np.float32) @ A
                    + 0.05 * rng.standard_normal((n, d)).astype(np.float32))
    io.write_fbin(str(tmp_path / "base.10M.fbin"), base)
    io.write_fbin(str(tmp_path / "query.10k.fbin"), mk(700))
    io.write_fbin(str(tmp_path / "query.train.10M.fbin"), mk(20000))
    common = [sys.executable, os.path.join(ROOT, "bench.py"), "--data-root", str(tmp_path), "--steps", "2", "--warmup", "1", "--nq", "512",
              "--gt-nq", "0", "--cpu-seconds", "0", "--sweep", "20,100", "--no-worstcase", "--no-fast", "--config1-nb", "0", "--configs", ""]
    common += ["--full-out", str(tmp_path / "full.json")]
    r = subprocess.run(common + ["--index-cache", str(tmp_path / "g.npz")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    _, d1 = _bench_records(r.stdout, str(tmp_path / "full.json"))
    assert d1["data"] == "files" and "genuine RoarGraph index built in the run" in d1["config"]["workload"]
    assert d1["L_pq_sweep"][-1]["recall_at_10"] > 0.9
    z = np.load(str(tmp_path / "g.npz"))
    io.write_index(str(tmp_path / "t2i_10M_roar.index"), z["off"], z["nbrs"], int(z["ep"]))
    r = subprocess.run(common, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    _, d2 = _bench_records(r.stdout, str(tmp_path / "full.json"))
    assert "index file t2i_10M_roar.index" in d2["config"]["workload"]
    assert [p["recall_at_10"] for p in d2["L_pq_sweep"]] == [p["recall_at_10"] for p in d1["L_pq_sweep"]]


@pytest.mark.parametrize("nd,d", [(1, 8), (300, 200), (4097, 200), (2500, 512), (70000, 24), (513, 104)])
def test_projection_ep_kernel_equals_host_loop(nd, d, oracle):
    """CalculateProjectionep (src/index_bipartite.cpp:2004-2041): the device form keeps the reference's summation orders
    (rows in index order per dimension, j order per row) and so returns the host loop's entry point -- including on sets
    with exactly tied distances (duplicated rows: the first index wins, :2031-2035) and with a large common offset, where
    a re-associated sum would round differently."""
    import ctypes as C
    import torch
    from roargraph_amd._lib import check, lib
    rng = np.random.default_rng(nd + d)
    base = (rng.standard_normal((nd, d)) * 3 + 100.0).astype(np.float32)
    if nd > 10:
        base[nd // 2:] = base[: nd - nd // 2]          # every row twice: ties everywhere
    host = C.c_uint32()
    check(lib().rg_projection_ep(base.ctypes.data_as(C.c_void_p), C.c_uint32(nd), C.c_uint32(d), C.c_uint32(d), C.byref(host)))
    bt = torch.from_numpy(base).cuda()
    devv = C.c_uint32()
    check(lib().rg_projection_ep_dev(C.c_void_p(bt.data_ptr()), C.c_uint32(nd), C.c_uint32(d), C.c_uint32(d), 0, C.byref(devv)))
    assert devv.value == oracle.projection_ep(base), "device entry point differs from the oracle's CalculateProjectionep"
    assert devv.value == host.value
    # an independent restatement of the same arithmetic in numpy (float32 throughout, sequential accumulation)
    if nd <= 5000:
        c = np.zeros(d, np.float32)
        for i in range(nd):
            c = (c + base[i]).astype(np.float32)
        c = (c / np.float32(nd)).astype(np.float32)
        diff = np.zeros(nd, np.float32)
        for j in range(d):
            t = (c[j] - base[:, j]).astype(np.float32)
            diff = (diff + (t * t).astype(np.float32)).astype(np.float32)
        assert int(np.argmin(diff)) == host.value


@pytest.mark.parametrize("dist_fn,metric,d", [("mips", "ip", 200), ("l2", "l2", 100)])
def test_compute_groundtruth_cli_multi_rank_leg(bins, oracle, tmp_path, dist_fn, metric, d):
    """`compute_groundtruth --devices 0,0,0`: the C++ CLI runs the multi-rank ground truth natively -- three ranks (threads)
    over row shards read straight from the .fbin file, query batches streamed (RG_GT_BATCH keeps them small), per-shard
    K-lists exchanged device to device, K3, rows written straight into the gt file.  d = 100 also exercises the zero
    padding of rows to the aligned stride on the way up."""
    from test_gpu_groundtruth import check_gt
    from roargraph_amd import synth
    base, q = synth.make_synth(88, 5003, 301, d)
    bf, qf, g1, g3 = (str(tmp_path / x) for x in ("b.fbin", "q.fbin", "gt1.bin", "gt3.bin"))
    io.write_fbin(bf, base); io.write_fbin(qf, q)
    env = dict(os.environ, RG_GT_BATCH="64")
    for devs, out in (("0", g1), ("0,0,0", g3)):
        r = subprocess.run([os.path.join(bins, "compute_groundtruth"), "--data_type", "float", "--dist_fn", dist_fn, "--base_file", bf,
                            "--query_file", qf, "--gt_file", out, "--K", "50", "--devices", devs], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
    i1, d1 = oracle.gt_load(g1)
    i3, d3 = oracle.gt_load(g3)
    assert (i1 == i3).all() and (d1.view(np.uint32) == d3.view(np.uint32)).all()
    ref_ids, _, ref_s = oracle.groundtruth_f64(base, q, metric, 50, nthreads=8)
    check_gt(base, q, metric, 50, i3, d3, ref_ids, ref_s)


def test_pinned_staged_host_transfers_round_trip():
    """synth.to_host / synth.to_device (bench.py's large transfers, round 5): gigabyte tensors cross PCIe through one pinned 64-MiB
    staging buffer instead of handing pageable memory to the runtime; values and dtypes survive, ragged last chunks included."""
    import torch
    from roargraph_amd import synth
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    t = torch.empty((700_001, 101), device=dev).normal_(generator=g)          # 283 MB: the staged path, last chunk ragged
    h = synth.to_host(t)
    assert h.dtype == np.float32 and h.shape == (700_001, 101) and (h == t.cpu().numpy()).all()
    back = synth.to_device(h, dev)
    assert back.dtype == torch.float32 and torch.equal(back, t)
    ids = torch.randint(0, 2 ** 31 - 1, (80_000_000,), dtype=torch.int32, device=dev, generator=g)     # 320 MB of int32
    assert (synth.to_host(ids) == ids.cpu().numpy()).all()
    small = torch.arange(10, device=dev)
    assert (synth.to_host(small) == np.arange(10)).all()
