"""-m gpu: concurrent searches on ONE index.

The reference calls SearchRoarGraph from many OpenMP threads against one index object
(tests/test_search_roargraph.cpp:203-209); its only shared mutable state is the visited-list pool behind a mutex
(include/visited_list_pool.h:47-65).  Here every launch-time buffer lives in a per-stream context handed out under the
index mutex, so host threads on distinct streams -- and several batches in flight on one stream -- must all return
the oracle's bits.
"""
import threading

import numpy as np
import pytest

from helpers import bits, small_set

pytestmark = pytest.mark.gpu


def _same(got, want):
    return ((got[0] == want[0]).all() and (bits(got[1]) == bits(want[1])).all() and (got[2] == want[2]).all()
            and (got[3] == want[3]).all())


@pytest.mark.parametrize("visited", [2, 0])
def test_two_threads_two_streams_device_form(oracle, visited):
    import torch
    from roargraph_amd.index import IndexBipartite
    base, q, off, nbrs, ep = small_set("ip", 4000, 200, nq=256)
    dev = torch.device("cuda", 0)
    ix = IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip")
    ix.set("visited", visited)
    ix.set("filter_log2", 6)          # forgetful filter: the id log / exact count path has work to do
    jobs = [(q[:128], 10, 100), (q[128:], 10, 500)]
    want = [oracle.search(base, "ip", off, nbrs, ep, qq, k, L, nthreads=4) for qq, k, L in jobs]
    errors = []

    def worker(t):
        try:
            qq, k, L = jobs[t]
            with torch.cuda.device(dev):
                s = torch.cuda.Stream(device=dev)
                qt = torch.from_numpy(qq).to(dev)
                nq = qq.shape[0]
                for it in range(12):
                    ids = torch.zeros((nq, k), dtype=torch.int32, device=dev); ds = torch.zeros((nq, k), device=dev)
                    cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
                    torch.cuda.synchronize()
                    ix.search_dev(qt, k, L, ids, ds, cm, hp, stream=s.cuda_stream)
                    ix.search_wait(s.cuda_stream)
                    got = (ids.cpu().numpy().view(np.uint32), ds.cpu().numpy(), cm.cpu().numpy().view(np.uint32),
                           hp.cpu().numpy().view(np.uint32))
                    if not _same(got, want[t]):
                        errors.append("thread %d iteration %d differs from the oracle" % (t, it))
                        return
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ix.close()
    assert not errors, errors


def test_many_threads_host_form(oracle):
    """The reference's pattern: T host threads call the search on one index object, each with its own queries."""
    from roargraph_amd.index import IndexBipartite
    base, q, off, nbrs, ep = small_set("l2", 3000, 200, nq=240)
    ix = IndexBipartite.from_arrays(base, off, nbrs, ep, metric="l2")
    want = oracle.search(base, "l2", off, nbrs, ep, q, 10, 100, nthreads=4)
    errors = []

    def worker(t):
        try:
            sl = slice(30 * t, 30 * t + 30)
            for _ in range(6):
                got = ix.SearchRoarGraph(q[sl], 10, 100)
                if not _same(got, tuple(w[sl] for w in want)):
                    errors.append("thread %d differs from the oracle" % t)
                    return
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ix.close()
    assert not errors, errors


def test_batches_in_flight_on_one_stream(oracle):
    """Five rg_search_dev calls (different beam widths, their own output buffers) before a single rg_search_wait: every
    batch is finished by the wait -- including the exact recount of id logs that overflowed (log_cap 64 forces it) --
    and a deferred "not enough results" of an early batch is still reported."""
    import torch
    from roargraph_amd._lib import RG_ERR_NOT_ENOUGH, RgError
    from roargraph_amd.index import IndexBipartite
    base, q, off, nbrs, ep = small_set("ip", 4000, 200, nq=96)
    dev = torch.device("cuda", 0)
    ix = IndexBipartite.from_arrays(base, off, nbrs, ep, metric="ip")
    ix.set("filter_log2", 6)
    ix.set("log_cap", 64)
    qt = torch.from_numpy(q).to(dev)
    outs = []
    for L in (20, 100, 50, 200, 10):
        ids = torch.zeros((96, 10), dtype=torch.int32, device=dev); ds = torch.zeros((96, 10), device=dev)
        cm = torch.zeros(96, dtype=torch.int32, device=dev); hp = torch.zeros(96, dtype=torch.int32, device=dev)
        ix.search_dev(qt, 10, L, ids, ds, cm, hp)
        outs.append((L, ids, ds, cm, hp))
    ix.search_wait()
    for L, ids, ds, cm, hp in outs:
        want = oracle.search(base, "ip", off, nbrs, ep, q, 10, L, nthreads=4)
        got = (ids.cpu().numpy().view(np.uint32), ds.cpu().numpy(), cm.cpu().numpy().view(np.uint32), hp.cpu().numpy().view(np.uint32))
        assert _same(got, want), L
    ix.close()
    # deferred error of the FIRST of three batches
    lonely = np.random.default_rng(0).standard_normal((50, 16)).astype(np.float32)
    ix = IndexBipartite.from_arrays(lonely, np.zeros(51, np.uint64), np.zeros(0, np.uint32), 3, metric="l2")
    qt = torch.from_numpy(lonely[:4].copy()).to(dev)
    bufs = [(torch.zeros((4, k), dtype=torch.int32, device=dev), torch.zeros((4, k), device=dev)) for k in (2, 1, 1)]
    for (ids, ds), k in zip(bufs, (2, 1, 1)):
        ix.search_dev(qt, k, 10, ids, ds)
    with pytest.raises(RgError, match="not enough results: 1, expected: 2") as e:
        ix.search_wait()
    assert e.value.code == RG_ERR_NOT_ENOUGH
    assert (bufs[1][0].cpu().numpy() == 3).all() and (bufs[2][0].cpu().numpy() == 3).all()
    ix.search_wait()      # nothing pending any more
    ix.close()


def test_fifty_opens_and_closes_reserve_no_new_address_space(oracle):
    """the balanced allocator uses a virtual range once (a runtime defect hands freed ranges back with stale
    translations), so a process that opens and closes indexes went through address space without bound and paid the placement probes
    on every open.  Round 5: a freed balanced buffer stays mapped in a cache and serves the next request of its size.  Fifty opens
    and closes of one index whose rows are a 3 GB balanced buffer: from the second open on no address space is reserved, no probe
    runs, every open is served from the cache, the placement is reported balanced, and the searches return the same bits; the
    cache goes back to the device on request."""
    import ctypes as C
    import os
    import torch
    from roargraph_amd._lib import check, lib
    from roargraph_amd.index import IndexBipartite
    if os.environ.get("RG_BALANCED_ALLOC") == "0":
        pytest.skip("the balanced allocator is switched off: there is no cache of mapped buffers to test")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(77)
    nb, d, deg, nq = 1_500_000, 512, 16, 256
    base = torch.empty((nb, d), device=dev).normal_(generator=g)
    nbrs = torch.randint(0, nb, (nb * deg,), dtype=torch.int32, device=dev, generator=g)
    off = torch.arange(0, nb + 1, dtype=torch.int64, device=dev) * deg
    q = torch.empty((nq, d), device=dev).normal_(generator=g)

    def stats():
        ex = (C.c_uint64 * 10)()
        check(lib().rg_mem_stats_ex(0, ex, 10))
        return [int(x) for x in ex]

    first = None
    va_after = []
    for it in range(50):
        ix = IndexBipartite.from_device(base, off, nbrs, 5, metric="l2")
        assert ix.stat("placement_balanced") == 1 and ix.stat("plain_allocs") == 0
        ids = torch.zeros((nq, 10), dtype=torch.int32, device=dev); ds = torch.zeros((nq, 10), device=dev)
        cm = torch.zeros(nq, dtype=torch.int32, device=dev); hp = torch.zeros(nq, dtype=torch.int32, device=dev)
        ix.search_dev(q, 10, 50, ids, ds, cm, hp); ix.search_wait()
        got = (ids.cpu().numpy(), ds.cpu().numpy().view(np.uint32), cm.cpu().numpy(), hp.cpu().numpy())
        if first is None:
            first = got
            ms = ix.mem_stats()
            assert ms["balanced_buffers"] >= 1 and ms["plain_fallbacks"] == 0 and ms["placement_balanced"]
        else:
            assert all((a == b).all() for a, b in zip(got, first)), it
        ix.close()
        va_after.append(stats())
    s1, s49 = va_after[1], va_after[49]
    assert s49[5] == s1[5], "address space reserved grew between the 2nd and the 50th open: %r -> %r" % (s1[5], s49[5])
    assert s49[3] == s1[3], "probes ran after the first opens"
    assert s49[6] - s1[6] >= 48, "opens were not served from the cache"
    assert s49[1] == 0 and s49[7] >= 3 * 2 ** 30            # nothing plain; the 3 GB buffer sits in the cache
    check(lib().rg_mem_release(0))
    assert stats()[7] == 0
    # and after the release the next open builds its buffer again (balanced; its addresses come from the arena reserved once per process --
    # round 6 -- so the address space reserved does not grow; RG_MEM_VA=leak, the mode of rounds 4 - 5, reserved a new range here)
    ix = IndexBipartite.from_device(base, off, nbrs, 5, metric="l2")
    assert ix.stat("placement_balanced") == 1 and stats()[5] >= s49[5] and stats()[0] > s49[0]
    ix.close()
    check(lib().rg_mem_release(0))


def test_open_search_close_reopen_200_iterations():
    """the lifecycle in which two bench runs of round 5 died of a GPU memory fault -- a large index opened, searched at narrow
    and wide beams in every exact form, closed, another shape opened in the memory the first one left (the allocator's cache of freed
    buffers, its remapped 1-GiB granules), the cache handed back every third time -- 200 times in one process.  A stale mapping shows as a GPU fault (the process dies) or as exact forms that disagree.
    (index_bipartite.h:27,62-64,105,133: the reference's lifecycle is constructor / load / search / destructor, any number of times.)"""
    import os
    from benchlib.stress import lifecycle_stress
    iters = int(os.environ.get("RG_STRESS_ITERS", "200"))
    # (host threads copying pageable memory meanwhile: RG_STRESS_HOST=N -- part of round 5's hypothesis, which round 6 ruled out; the default run
    # does without them: they triple the test's time through the interpreter lock)
    r = lifecycle_stress(iters, scale=float(os.environ.get("RG_STRESS_SCALE", "0.8")), host_load=int(os.environ.get("RG_STRESS_HOST", "0")))
    assert r["iterations"] == iters


def test_kill_switch_of_the_balanced_allocator():
    """RG_BALANCED_ALLOC=0 must leave a working library that never touches the virtual-memory API -- every large buffer
    a plain hipMalloc, no probe launched, no address space reserved -- through the same lifecycle (its own process: the switch is read once)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, ctypes as C; sys.path.insert(0, %r)\n"
            "from benchlib.stress import lifecycle_stress\n"
            "from roargraph_amd._lib import lib\n"
            "r = lifecycle_stress(6, scale=0.8, host_load=0)\n"
            "ex = (C.c_uint64 * 10)(); lib().rg_mem_stats_ex(0, ex, 10)\n"
            "print('RESULT ' + json.dumps({'iterations': r['iterations'], 'balanced': int(ex[0]), 'probes': int(ex[3]), 'va': int(ex[5]), 'cached': int(ex[7])}))\n" % root)
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RG_BALANCED_ALLOC="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert r == {"iterations": 6, "balanced": 0, "probes": 0, "va": 0, "cached": 0}, r


def test_allocator_walk_under_address_churn():
    """The root cause of the GPU memory faults of rounds 5 and 6, as far as user space can see it: the runtime hands hipMemAddressReserve
    address ranges that plain hipMalloc'ed buffers occupied moments earlier, and the FIRST TOUCH of a granule mapped into such a range can
    fault (scripts/r06/walk_stress.py with RG_MEM_VA=leak, the address policy of rounds 4 - 5: 5 of 5 runs died within 2,100 granules,
    `Memory access fault by GPU` on a 1-GiB-aligned granule just mapped -- profiles/r06/walk_stress_summary.txt; with the arena of round 6 --
    every address carved once from one range reserved when the pool is first used -- 0 faults in 57,024 granules).  Here: the same loop in
    the default (arena) mode for 30 s -- torch and plain allocations churned between rounds of 48 granules created, mapped, zeroed, probed
    and dropped.  A regression of the address policy kills the process."""
    import ctypes as C
    import os
    import time
    import torch
    from roargraph_amd._lib import check, lib
    if os.environ.get("RG_BALANCED_ALLOC") == "0":
        pytest.skip("the balanced allocator is switched off")
    assert os.environ.get("RG_MEM_VA", "arena") == "arena", "this test must not run in the address modes that fault"
    dev = torch.device("cuda", 0)
    n = C.c_uint64()
    total = 0
    t0 = time.time()
    while time.time() - t0 < float(os.environ.get("RG_WALK_SECONDS", "30")):
        keep = [torch.empty((int(s * 2 ** 30) // 4,), dtype=torch.float32, device=dev).fill_(1.0) for s in (0.3, 1.7, 4.0, 0.05, 9.0, 2.5, 0.6)]
        torch.cuda.synchronize()
        del keep
        torch.cuda.empty_cache()
        check(lib().rg_mem_walk_stress(0, 48, 1, C.byref(n)))
        total += n.value
    assert total >= 48
    check(lib().rg_mem_release(0))
