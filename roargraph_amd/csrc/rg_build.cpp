// rg_build.cpp -- RoarGraph's graph construction (a "next" row).  It consumes the ground truth
// produced by K2 and emits the .index the search path loads, which closes the pipeline  GT (GPU) -> build -> search (GPU).
// rg_build_roargraph is the all-CPU restatement (one thread = the reference's sequence); rg_build_roargraph_gpu moves what
// dominates the reference's build time to the GPU: the entry point, the n beam searches of phase 3 (K1 in build mode) and
// the occlusion pruning of their expansion lists (rg_build_prune.hip) -- the host threads keep phases 1, 2, 4, 5 and the
// reverse-edge insertion of phase 3.
//
// Follows IndexBipartite::BuildRoarGraph (src/index_bipartite.cpp:143-233):
//   CalculateProjectionep                       :2004-2041   entry point = base row nearest to the centroid
//   LinkProjection                              :1043-1277   phases 1-5 below
//   PruneBiSearchBaseGetBase                    :1612-1694
//   ProjectionAddReverse / PruneProjectionReverseCandidates             :1391-1432 / :1526-1610
//   SearchProjectionGraphInternal               :1279-1350
//   PruneProjectionBaseSearchCandidates         :1846-1940
//   SupplyAddReverse / PruneProjectionInternalReverseCandidates         :1352-1389 / :1434-1524
//
// PARITY STATUS: unpinned.  The reference translation unit cannot be compiled in this image (Boost / tsl headers are
// absent), so there is no reference-built .index to compare with; the code below is a line-by-line behavioural
// restatement, including the reference's quirks that shape the output (e.g. the zero-initialised phantom entries of
// PruneProjectionInternalReverseCandidates, :1438), and is validated by properties (degree bounds, determinism at one
// thread, recall of the search over the built index).  Out-of-range accesses the reference would perform on empty
// candidate pools are guarded instead of replicated.
//
// Threads: one thread reproduces the reference's T=1 sequence.  More threads do NOT race on the lists as the reference's do
// (its multi-threaded result depends on scheduling): phases 1 and 2 are replayed list by list in the one-thread
// order (phase1_replay, phase2_windows), phase 3 runs in fixed batches over a frozen graph and links them in node order
// (link_batch) -- one result for any thread count, on the host or with the GPU, equal to the oracle's restatement given
// the same batch list (rg_build_schedule).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <exception>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <climits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <sched.h>

#include <hip/hip_runtime.h>

#include "rg.h"
#include "rg_index_struct.h"
#include "rg_internal.h"
#include "rg_mem.h"

namespace rg {
namespace {

// ---- distance on the host, same value as the reference's AVX-512 kernels (distance.h:39-87, 180-223) ----------------
// lane j of a 16-wide accumulator takes elements j, j+16, ... with one FMA each; folds 16->8 (+8-wide tail) ->4 -> 2 hadds.
__attribute__((target("fma"))) float fold16(const float (&acc)[16], const float *a, const float *b, unsigned rem, bool l2) {
    float s8[8], s4[4];
    for (int j = 0; j < 8; ++j) s8[j] = acc[j + 8] + acc[j];
    if (rem >= 8) {
        for (int j = 0; j < 8; ++j) {
            if (l2) { const float t = a[j] - b[j]; s8[j] = __builtin_fmaf(t, t, s8[j]); }
            else s8[j] = __builtin_fmaf(a[j], b[j], s8[j]);
        }
    }
    for (int j = 0; j < 4; ++j) s4[j] = s8[j + 4] + s8[j];
    return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}
__attribute__((target("fma"))) float dist_scalar(const float *a, const float *b, unsigned d, bool l2) {
    float acc[16] = {0};
    unsigned i = 0;
    for (; d - i >= 16; i += 16)
        for (int j = 0; j < 16; ++j) {
            if (l2) { const float t = a[i + j] - b[i + j]; acc[j] = __builtin_fmaf(t, t, acc[j]); }
            else acc[j] = __builtin_fmaf(a[i + j], b[i + j], acc[j]);
        }
    const float r = fold16(acc, a + i, b + i, d - i, l2);
    return l2 ? r : -r;
}
#if defined(__x86_64__)
__attribute__((target("avx512f,fma"))) float dist_avx512(const float *a, const float *b, unsigned d, bool l2) {
    __m512 acc = _mm512_setzero_ps();
    unsigned i = 0;
    if (l2) {
        for (; d - i >= 16; i += 16) {
            const __m512 t = _mm512_sub_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i));
            acc = _mm512_fmadd_ps(t, t, acc);
        }
    } else {
        for (; d - i >= 16; i += 16) acc = _mm512_fmadd_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i), acc);
    }
    float lanes[16];
    _mm512_storeu_ps(lanes, acc);
    const float r = fold16(lanes, a + i, b + i, d - i, l2);
    return l2 ? r : -r;
}
#endif

struct Nb {
    uint32_t id;
    float dist;
    bool operator<(const Nb &o) const { return dist < o.dist || (dist == o.dist && id < o.id); }  // neighbor.h:29-31
};

// bounded sorted queue with the reference's insert / closest_unexpanded rules (neighbor.h:150-192)
struct Beam {
    struct E { uint32_t id; float dist; bool done; };
    std::vector<E> e;
    size_t size = 0, cap = 0, cur = 0;
    void reset(size_t c) { cap = c; e.resize(c + 1); size = 0; cur = 0; }
    static bool less(uint32_t ia, float da, uint32_t ib, float db) { return da < db || (da == db && ia < ib); }
    void insert(uint32_t id, float d) {
        if (size == cap && less(e[size - 1].id, e[size - 1].dist, id, d)) return;
        size_t lo = 0, hi = size;
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if (less(id, d, e[mid].id, e[mid].dist)) hi = mid;
            else if (e[mid].id == id) return;
            else lo = mid + 1;
        }
        if (lo < cap) std::memmove(&e[lo + 1], &e[lo], (size - lo) * sizeof(E));
        e[lo] = {id, d, false};
        if (size < cap) ++size;
        if (lo < cur) cur = lo;
    }
    bool has_open() const { return cur < size; }
    E pop() {
        e[cur].done = true;
        const size_t pre = cur;
        while (cur < size && e[cur].done) ++cur;
        return e[pre];
    }
};

struct Builder {
    const float *base;
    size_t stride;
    unsigned dim;
    uint32_t nd;
    bool l2, avx512;
    uint32_t M, L, Nq;  // M_pjbp, L_pjpq, M_sq
    uint32_t ep = 0;
    bool ep_known = false;
    float *d_base_pre = nullptr;   // GPU-assisted build: the base goes up once, for the entry point and for phase 3
    std::vector<std::vector<uint32_t>> proj, supply;
    std::vector<uint8_t> dirty;    // GPU phase 3: supply rows changed since the graph snapshot on the device was refreshed
    std::vector<std::mutex> locks;
    int threads = 1;

    float cmp(uint32_t a, uint32_t b) const {
        const float *pa = base + (size_t)a * stride, *pb = base + (size_t)b * stride;
#if defined(__x86_64__)
        if (avx512) return dist_avx512(pa, pb, dim, l2);
#endif
        return dist_scalar(pa, pb, dim, l2);
    }
    static bool has(const std::vector<uint32_t> &v, uint32_t x) { return std::find(v.begin(), v.end(), x) != v.end(); }

    // An exception on a worker thread (bad_alloc from a list that grows, a vector of a pruning rule) would reach
    // std::terminate on a std::thread, and a worker that died would leave its team mates in the barrier.  Workers catch; the
    // first exception is kept, the others are told to stop (the loop's counter is pushed past its end, the team's barrier is
    // poisoned), and the caller's thread rethrows after the join -- build_impl turns it into a status.
    // RG_BUILD_FAULT=<n> (tests only): the n-th multi-threaded region of a build throws std::bad_alloc on one of its worker threads
    static bool fault_here() {
        static const long at = getenv("RG_BUILD_FAULT") ? atol(getenv("RG_BUILD_FAULT")) : 0;
        static std::atomic<long> region(0);
        return at > 0 && region.fetch_add(1) + 1 == at;
    }
    struct FirstError {
        std::mutex mu;
        std::exception_ptr ep;
        void keep(std::exception_ptr e) { std::lock_guard<std::mutex> lk(mu); if (!ep) ep = e; }
        void rethrow() { if (ep) std::rethrow_exception(ep); }
    };
    // static-chunk (phases 1, 2) or dynamic-chunk (phases 3-5) loops, as the reference's omp schedules
    void parallel_for(uint32_t n, uint32_t chunk, const std::function<void(uint32_t, int)> &fn) {
        if (threads <= 1) { for (uint32_t i = 0; i < n; ++i) fn(i, 0); return; }
        std::atomic<uint32_t> next(0);
        std::atomic<bool> stop(false);
        FirstError err;
        const bool fault = fault_here();
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                try {
                    if (fault && t == threads - 1) throw std::bad_alloc();
                    for (;;) {
                        const uint32_t lo = next.fetch_add(chunk);
                        if (lo >= n || stop.load(std::memory_order_relaxed)) break;
                        const uint32_t hi = std::min(n, lo + chunk);
                        for (uint32_t i = lo; i < hi; ++i) fn(i, t);
                    }
                } catch (...) {
                    err.keep(std::current_exception());
                    stop.store(true, std::memory_order_relaxed);
                }
            });
        for (auto &th : pool) th.join();
        err.rethrow();
    }

    // T persistent threads running fn(t, team) with a spinning barrier between their steps (phase 2 below has ~10^4 steps of
    // a few microseconds each: thread creation per step would cost more than the steps)
    struct TeamAborted {};     // thrown by barrier() once a team mate has failed
    struct Team {
        int T;
        std::atomic<int> arrived{0}, sense{0}, aborted{0};
        explicit Team(int t) : T(t) {}
        void abort() {
            aborted.store(1, std::memory_order_release);
            syscall(SYS_futex, reinterpret_cast<int *>(&sense), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
        }
        void barrier(int &local) {
            local ^= 1;
            if (aborted.load(std::memory_order_acquire)) throw TeamAborted();
            if (arrived.fetch_add(1, std::memory_order_acq_rel) == T - 1) {
                arrived.store(0, std::memory_order_relaxed);
                sense.store(local, std::memory_order_release);
                if (T > 1) syscall(SYS_futex, reinterpret_cast<int *>(&sense), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
            } else {
                // a short spin, then sleep in the kernel: the threads may outnumber the CPUs this process may use (a CPU quota
                // makes spinning waiters starve the thread everybody waits for)
                for (unsigned spins = 0; spins < 64 && sense.load(std::memory_order_acquire) != local; ++spins) {
#if defined(__x86_64__)
                    _mm_pause();
#endif
                }
                while (sense.load(std::memory_order_acquire) != local) {
                    if (aborted.load(std::memory_order_acquire)) throw TeamAborted();
                    // (a bounded sleep: a team mate's abort() may land between the check above and the wait)
                    struct timespec ts = {0, 2000000};
                    syscall(SYS_futex, reinterpret_cast<int *>(&sense), FUTEX_WAIT_PRIVATE, local ^ 1, &ts, nullptr, 0);
                }
            }
        }
    };
    static_assert(sizeof(std::atomic<int>) == sizeof(int), "the futex word is the atomic itself");
    void run_team(const std::function<void(int, Team &)> &fn) {
        Team team(std::max(1, threads));
        if (team.T == 1) { fn(0, team); return; }
        FirstError err;
        const bool fault = fault_here();
        std::vector<std::thread> pool;
        for (int t = 0; t < team.T; ++t)
            pool.emplace_back([&, t] {
                try {
                    if (fault && t == team.T - 1) throw std::bad_alloc();
                    fn(t, team);
                } catch (const TeamAborted &) {
                    // a team mate failed: its exception is the one reported
                } catch (...) {
                    err.keep(std::current_exception());
                    team.abort();
                }
            });
        for (auto &th : pool) th.join();
        err.rethrow();
    }
    // which thread applies the reverse edges into list x: any fixed map gives the same graph, every list has ONE writer
    static uint32_t owner_of(uint32_t x, uint32_t T) { return (uint32_t)(((uint64_t)(x * 2654435761u) * T) >> 32); }

    // occlusion scan shared by all prune routines: p survives if it is not yet chosen and no chosen r is closer to it
    // than p is to the pivot
    // the pruning rules evaluate distances between base rows picked by id: on a 10M-row base every one of them is a DRAM
    // miss, so the rows a rule is about to touch are requested ahead of the arithmetic
    void prefetch_row(uint32_t id) const {
        const char *r = reinterpret_cast<const char *>(base + (size_t)id * stride);
        for (size_t o = 0; o < (size_t)dim * 4; o += 64) __builtin_prefetch(r + o, 0, 3);
    }

    // Monotone: `result` only grows, so a candidate found occluded once stays occluded, and a candidate already in
    // `result` is "occluded" by itself (the id test).  The reference's second sweeps (:1662-1676, :1576-1590, :1496-1510,
    // :1912-1926) walk candidates the first sweep has already decided -- each iteration re-derives "occluded" and adds
    // nothing.  Those iterations are skipped below; the lists produced are the same (checked at one thread against the
    // sweeps written out, scripts/exp/build_t1_hashes.py), the distance evaluations are halved.
    bool occluded(const Nb &p, const std::vector<uint32_t> &result) const {
        for (uint32_t r : result) {
            if (p.id == r) return true;
            if (cmp(p.id, r) < p.dist) return true;
        }
        return false;
    }

    // :1612-1694
    void prune_get_base(std::vector<Nb> &pool, uint32_t tgt, std::vector<uint32_t> &out) const {
        std::vector<Nb> bp;
        std::vector<uint32_t> ids;
        for (const Nb &b : pool)
            if (!has(ids, b.id)) {
                if (b.id == tgt) continue;
                bp.push_back(b);
                ids.push_back(b.id);
            }
        std::vector<uint32_t> result;
        if (bp.empty()) { out = result; return; }
        std::sort(bp.begin(), bp.end());
        result.reserve(2 * M);
        uint32_t start = 0;
        result.push_back(bp[0].id);
        while (result.size() < M && (++start) < bp.size()) {
            const Nb &p = bp[start];
            if (!occluded(p, result) && p.id != tgt) result.push_back(p.id);
        }
        // second sweep over the unsorted pool (:1662-1676): every pool entry is the target, or bp[0] (in `result`), or was
        // decided by the first sweep, which ran to the end of bp if `result` is still short -- a no-op, skipped
        for (size_t i = 1; i < bp.size() && result.size() < M; ++i)
            if (!has(result, bp[i].id) && bp[i].id != tgt) result.push_back(bp[i].id);
        out = result;
    }

    // :1526-1610 (phantoms = false) and :1434-1524 (phantoms = true: the queue starts with list.size() zero entries)
    void prune_reverse(uint32_t src, std::vector<uint32_t> &list, bool phantoms) const {
        std::vector<Nb> pq;
        if (phantoms) pq.assign(list.size(), Nb{0u, 0.0f});
        for (uint32_t id : list) prefetch_row(id);
        for (uint32_t id : list) {
            const float d = cmp(src, id);
            bool seen = false;
            for (const Nb &x : pq) if (x.id == id) { seen = true; break; }
            if (!seen) pq.push_back(Nb{id, d});
        }
        std::vector<uint32_t> result;
        if (pq.empty()) { list = result; return; }
        std::sort(pq.begin(), pq.end());
        result.reserve(2 * M);
        uint32_t start = 0;
        if (pq[start].id == src) ++start;
        if (start >= pq.size()) { list = result; return; }
        result.push_back(pq[start].id);
        while (result.size() < M && (++start) < pq.size()) {
            const Nb &p = pq[start];
            if (!occluded(p, result) && p.id != src) result.push_back(p.id);
        }
        // second sweep (:1576-1590 / :1496-1510) over pq[1 ..): all of them in `result` or decided by the first sweep -- skipped
        if (!phantoms)   // :1594-1598 top-up in the original list order
            for (size_t i = 0; i < list.size() && result.size() < M; ++i)
                if (!has(result, list[i])) result.push_back(list[i]);
        list = result;
    }

    // :1391-1432 (on proj, limit M, plain prune) and :1352-1389 (on supply, limit 2M, phantom prune)
    void add_reverse(std::vector<std::vector<uint32_t>> &g, uint32_t src, uint32_t limit, bool phantoms) {
        std::vector<uint32_t> nbrs;
        {   // snapshot (identical to iterating the live list at one thread: pruning `des` never rewrites g[src])
            std::lock_guard<std::mutex> guard(locks[src]);
            nbrs = g[src];
        }
        for (size_t i = 0; i < nbrs.size(); ++i) {
            const uint32_t des = nbrs[i];
            std::vector<uint32_t> copy;
            bool need = false;
            {
                std::lock_guard<std::mutex> guard(locks[des]);
                std::vector<uint32_t> &dn = g[des];
                if (has(dn, src)) continue;
                if (dn.size() < limit) dn.push_back(src);
                else { need = true; copy = dn; }
            }
            if (need) {
                copy.push_back(src);
                prune_reverse(des, copy, phantoms);
                std::lock_guard<std::mutex> guard(locks[des]);
                g[des] = copy;
            }
        }
    }

    // prune_reverse (plain rule, :1526-1610) for a list that is pruned again and again -- phase 2 keeps the lists of popular base
    // points at their bound, so EVERY further insertion re-prunes 36 candidates of which 35 were there the last time.  The
    // cache keeps, for the first k entries of the list as it stands, their distance to the list's owner and to each other
    // (the rule's `djk`), so a call computes only what involves the new entry; every value it uses is the value cmp()
    // returns (cmp is symmetric bit for bit: a*b and (a-b)^2 commute), so the list that comes out is prune_reverse's.
    // Layout: k, ids[cap], dx[cap], pair[cap][cap] as 32-bit words; kUnk marks a distance not computed yet (a genuine
    // NaN of that pattern would just be recomputed every time).
    static constexpr uint32_t kUnk = 0x7fc0dead;
    static float unk() { float f; const uint32_t u = kUnk; std::memcpy(&f, &u, 4); return f; }
    static bool is_unk(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u == kUnk; }
    size_t pcache_words() const { const size_t cap = M + 1; return 1 + cap + cap + cap * cap; }
    void prune_reverse_cached(uint32_t des, std::vector<uint32_t> &list, uint32_t *c, std::vector<float> &scratch) const {
        const uint32_t cap = M + 1, m = (uint32_t)list.size();
        if (m > cap) { c[0] = 0; prune_reverse(des, list, false); return; }
        uint32_t *ids = c + 1;
        float *dx = reinterpret_cast<float *>(c + 1 + cap), *pair = dx + cap;
        uint32_t k = c[0];
        if (k > m || std::memcmp(ids, list.data(), (size_t)k * 4) != 0) k = 0;
        const float U = unk();
        for (uint32_t i = k; i < m; ++i) {
            ids[i] = list[i];
            dx[i] = U;
            for (uint32_t j = 0; j < m; ++j) pair[(size_t)i * cap + j] = pair[(size_t)j * cap + i] = U;
        }
        struct E { uint32_t id; float dist; uint32_t idx; };
        E pq[64 + 8];
        std::vector<E> big;
        E *q = pq;
        if (m > 64) { big.resize(m); q = big.data(); }
        uint32_t n = 0;
        for (uint32_t i = 0; i < m; ++i) {
            if (is_unk(dx[i])) { prefetch_row(ids[i]); }
        }
        for (uint32_t i = 0; i < m; ++i) {
            if (is_unk(dx[i])) dx[i] = cmp(des, ids[i]);
            bool seen = false;
            for (uint32_t j = 0; j < n; ++j) if (q[j].id == ids[i]) { seen = true; break; }
            if (!seen) q[n++] = E{ids[i], dx[i], i};
        }
        uint32_t res[64 + 8];
        std::vector<uint32_t> bigres;
        uint32_t *r = res;
        if (m > 64) { bigres.resize(m); r = bigres.data(); }
        uint32_t nres = 0;
        if (n) {
            std::sort(q, q + n, [](const E &a, const E &b) { return a.dist < b.dist || (a.dist == b.dist && a.id < b.id); });
            uint32_t start = 0;
            if (q[start].id == des) ++start;
            if (start < n) {
                r[nres++] = q[start].idx;
                while (nres < M && (++start) < n) {
                    const E &p = q[start];
                    bool occ = false;
                    for (uint32_t t = 0; t < nres && !occ; ++t) {
                        if (ids[r[t]] == p.id) { occ = true; break; }
                        float &d = pair[(size_t)p.idx * cap + r[t]];
                        if (is_unk(d)) { d = cmp(p.id, ids[r[t]]); pair[(size_t)r[t] * cap + p.idx] = d; }
                        if (d < p.dist) occ = true;
                    }
                    if (!occ && p.id != des) r[nres++] = p.idx;
                }
                for (uint32_t i = 0; i < m && nres < M; ++i) {          // :1594-1598 top-up in the original list order
                    bool in = false;
                    for (uint32_t t = 0; t < nres; ++t) if (ids[r[t]] == ids[i]) { in = true; break; }
                    if (!in) r[nres++] = i;
                }
            }
        }
        // the list that comes out, and the distances among its members carried over in its order
        scratch.resize((size_t)nres * nres + 2 * (size_t)nres);
        float *ndx = scratch.data(), *npair = ndx + nres;
        uint32_t *nid = reinterpret_cast<uint32_t *>(npair + (size_t)nres * nres);
        for (uint32_t a = 0; a < nres; ++a) {
            nid[a] = ids[r[a]];
            ndx[a] = dx[r[a]];
            for (uint32_t b = 0; b < nres; ++b) npair[(size_t)a * nres + b] = pair[(size_t)r[a] * cap + r[b]];
        }
        list.assign(nid, nid + nres);
        std::memcpy(ids, nid, (size_t)nres * 4);
        std::memcpy(dx, ndx, (size_t)nres * 4);
        for (uint32_t a = 0; a < nres; ++a) std::memcpy(pair + (size_t)a * cap, npair + (size_t)a * nres, (size_t)nres * 4);
        c[0] = nres;
    }

    // one step of add_reverse's loop on a list this thread owns (no lock: deterministic schedules give every list one writer)
    void apply_insert(std::vector<uint32_t> &dn, uint32_t des, uint32_t src, uint32_t limit, bool phantoms) const {
        if (has(dn, src)) return;
        dn.push_back(src);
        if (dn.size() > limit) prune_reverse(des, dn, phantoms);
    }

    // ---- deterministic multi-threaded forms of the three places where the reference's threads race on the lists ----------
    // All three rest on one observation: a reverse-edge insertion (src -> des) reads and writes the list of `des` ONLY
    // (prune_reverse scores des against its own list members).  So once it is known WHICH insertions reach a list and in what
    // order, every list can be replayed on its own by one thread, and the outcome is the one-thread outcome -- for any
    // number of threads, without locks.
    //
    // Phase 1 (:1059-1097).  The pruned list of training query sq depends on its knn row only (`lists`: [nq][M+1], length +
    // ids, complete), so the whole event sequence of a list x is static: a WRITE by every query whose nearest base point
    // is x, an INSERT (of that query's nearest point) by every query whose pruned list holds x, in query order.  A WRITE
    // replaces the list, so x starts from its LAST write and replays the inserts of the later queries.
    void phase1_replay(const uint32_t *knn, uint32_t nq, uint32_t kdim, const std::vector<uint32_t> &lists) {
        const uint32_t W = M + 1, none = 0xffffffffu;
        std::vector<uint32_t> last(nd, none);
        for (uint32_t sq = 0; sq < nq; ++sq) last[knn[(size_t)sq * kdim]] = sq;
        std::vector<uint32_t> cnt(nd, 0);
        auto live = [&](uint32_t des, uint32_t sq) { return last[des] == none || sq > last[des]; };
        parallel_for(nq, 4096, [&](uint32_t sq, int) {
            const uint32_t *l = lists.data() + (size_t)sq * W;
            for (uint32_t i = 0; i < l[0]; ++i)
                if (live(l[1 + i], sq)) __atomic_fetch_add(&cnt[l[1 + i]], 1u, __ATOMIC_RELAXED);
        });
        std::vector<uint64_t> off((size_t)nd + 1, 0);
        for (uint32_t x = 0; x < nd; ++x) { off[x + 1] = off[x] + cnt[x]; cnt[x] = 0; }
        std::vector<uint32_t> ev(off[nd]);
        parallel_for(nq, 4096, [&](uint32_t sq, int) {
            const uint32_t *l = lists.data() + (size_t)sq * W;
            for (uint32_t i = 0; i < l[0]; ++i) {
                const uint32_t des = l[1 + i];
                if (live(des, sq)) ev[off[des] + __atomic_fetch_add(&cnt[des], 1u, __ATOMIC_RELAXED)] = sq;
            }
        });
        auto replay = [&](uint32_t x) {
            if (last[x] == none && cnt[x] == 0) return;
            std::vector<uint32_t> &dn = proj[x];
            dn.clear();
            if (last[x] != none) {
                const uint32_t *l = lists.data() + (size_t)last[x] * W;
                dn.assign(l + 1, l + 1 + l[0]);
            }
            uint32_t *e = ev.data() + off[x];
            std::sort(e, e + cnt[x]);                         // query order (the fill above ran in parallel)
            for (uint32_t i = 0; i < cnt[x]; ++i) apply_insert(dn, x, knn[(size_t)e[i] * kdim], M, false);
        };
        // the longest replays first (hubs take tens of thousands of inserts, one after another), then everything else
        std::vector<uint32_t> heavy;
        for (uint32_t x = 0; x < nd; ++x) if (cnt[x] >= 4096) heavy.push_back(x);
        std::sort(heavy.begin(), heavy.end(), [&](uint32_t a, uint32_t b) { return cnt[a] > cnt[b] || (cnt[a] == cnt[b] && a < b); });
        std::atomic<uint32_t> next_heavy(0), next_light(0);
        if (timing) fprintf(stderr, "[rg_build]   phase 1: %zu reverse edges to replay, %zu lists with >= 4096 of them (the longest %u)\n",
                            (size_t)off[nd], heavy.size(), heavy.empty() ? 0u : cnt[heavy[0]]);
        run_team([&](int, Team &) {
            for (;;) {
                const uint32_t i = next_heavy.fetch_add(1);
                if (i >= heavy.size()) break;
                replay(heavy[i]);
            }
            for (;;) {
                const uint32_t lo = next_light.fetch_add(256);
                if (lo >= nd) break;
                for (uint32_t x = lo; x < std::min(nd, lo + 256); ++x)
                    if (cnt[x] < 4096) replay(x);
            }
        });
    }

    // Phase 2 (:1100-1107: for every node in order, insert it into the lists of its neighbours).  Node j reads its own list
    // when its turn comes, and earlier turns may have changed that list -- but only turns of nodes i < j that held j in
    // THEIR list, and (j being later) they can only hold j if they held it when the phase began.  So consecutive nodes
    // are cut into windows in which no node is listed by an earlier node of the same window (pred[j] = the last i < j whose
    // starting list holds j; j opens a new window if pred[j] is inside the current one): inside a window every node's list is
    // still what it was when the window began, all their insertions are known, and each target list replays its own in
    // source order -- the one-thread result exactly.
    void phase2_windows() {
        std::vector<uint32_t> pred(nd, 0);                // pred[j] = 1 + the last i < j whose list holds j (0: none)
        parallel_for(nd, 4096, [&](uint32_t i, int) {
            for (uint32_t j : proj[i])
                if (j > i) {
                    uint32_t cur = __atomic_load_n(&pred[j], __ATOMIC_RELAXED);
                    while (cur < i + 1 && !__atomic_compare_exchange_n(&pred[j], &cur, i + 1, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
                }
        });
        std::vector<uint32_t> wend;                       // window ends
        size_t maxw = 0;
        for (uint32_t a = 0, j = 0; j <= nd; ++j)
            if (j == nd || (pred[j] != 0 && pred[j] - 1 >= a)) {
                if (j > a) { wend.push_back(j); maxw = std::max<size_t>(maxw, j - a); }
                a = j;
                if (j == nd) break;
            }
        std::vector<uint32_t>().swap(pred);
        lap("phase 2 windows");
        if (timing) fprintf(stderr, "[rg_build]   phase 2: %zu windows of independent nodes (mean %.0f, longest %zu)\n", wend.size(),
                            wend.empty() ? 0.0 : (double)nd / wend.size(), maxw);
        std::vector<uint32_t> ev_des(maxw * (size_t)M + 1), ev_src(maxw * (size_t)M + 1);
        std::atomic<long long> st_events{0}, st_present{0}, st_pruned{0}, st_ns_apply{0}, st_ns_total{0};
        // distance caches of the lists that get pruned (prune_reverse_cached), each touched by its owner thread only; at most
        // 1.5 GB of them, first come first served (popular lists fill up first); RG_BUILD_NO_PRUNE_CACHE=1 turns them off
        std::vector<uint32_t *> pc(nd, nullptr);
        const size_t pc_words = pcache_words();
        std::atomic<long long> pc_left{getenv("RG_BUILD_NO_PRUNE_CACHE") ? 0 : (long long)(1500000000ull / (pc_words * 4))};
        auto insert_cached = [&](std::vector<uint32_t> &dn, uint32_t des, uint32_t src, std::vector<float> &scratch) {
            if (has(dn, src)) return;
            dn.push_back(src);
            if (dn.size() <= M) return;
            uint32_t *c = pc[des];
            if (!c && pc_left.load(std::memory_order_relaxed) > 0 && pc_left.fetch_sub(1) > 0) {
                c = pc[des] = new uint32_t[pc_words];
                c[0] = 0;
            }
            if (c) prune_reverse_cached(des, dn, c, scratch);
            else prune_reverse(des, dn, false);
        };
        run_team([&](int t, Team &team) {
            int sense = 0;
            const uint32_t T = (uint32_t)team.T;
            std::vector<uint32_t> pre(maxw + 1);          // where each node's insertions start in the window's event list
            uint32_t a = 0;
            long long my_events = 0, my_present = 0, my_pruned = 0, my_ns = 0;
            std::vector<float> scratch;
            const auto t_begin = std::chrono::steady_clock::now();
            for (size_t w = 0; w < wend.size(); ++w) {
                const uint32_t b = wend[w], n = b - a;
                pre[0] = 0;
                for (uint32_t i = 0; i < n; ++i) pre[i + 1] = pre[i] + (uint32_t)proj[a + i].size();
                for (uint32_t i = (uint32_t)t; i < n; i += T) {
                    const std::vector<uint32_t> &l = proj[a + i];
                    for (size_t k = 0; k < l.size(); ++k) { ev_des[pre[i] + k] = l[k]; ev_src[pre[i] + k] = a + i; }
                }
                team.barrier(sense);
                const uint32_t ne = pre[n];
                if (!timing) {
                    for (uint32_t e = 0; e < ne; ++e)
                        if (owner_of(ev_des[e], T) == (uint32_t)t) insert_cached(proj[ev_des[e]], ev_des[e], ev_src[e], scratch);
                } else {
                    const auto t0 = std::chrono::steady_clock::now();
                    for (uint32_t e = 0; e < ne; ++e)
                        if (owner_of(ev_des[e], T) == (uint32_t)t) {
                            std::vector<uint32_t> &dn = proj[ev_des[e]];
                            ++my_events;
                            if (has(dn, ev_src[e])) { ++my_present; continue; }
                            if (dn.size() >= M) ++my_pruned;
                            insert_cached(dn, ev_des[e], ev_src[e], scratch);
                        }
                    my_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                }
                team.barrier(sense);
                a = b;
            }
            if (timing) {
                st_events += my_events; st_present += my_present; st_pruned += my_pruned; st_ns_apply += my_ns;
                st_ns_total += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
            }
        });
        size_t ncache = 0;
        for (uint32_t *c : pc) if (c) { ++ncache; delete[] c; }
        if (timing) fprintf(stderr, "[rg_build]   phase 2: %lld insertions tried, %lld already present, %lld into a full list (pruned; %zu lists with a distance cache); thread time applying %.1f s of %.1f s\n",
                            st_events.load(), st_present.load(), st_pruned.load(), ncache, st_ns_apply.load() * 1e-9, st_ns_total.load() * 1e-9);
    }

    // Linking a batch of phase 3 (:1209-1215 for nodes [b0, b0 + n), whose pruned lists `lists` ([n][M+1]) were all computed
    // from the graph as it stood before the batch): node x's list is WRITTEN at its turn and takes INSERTs of the nodes whose
    // pruned list holds x, in node order -- inserts of earlier nodes of the same batch into x are overwritten by x's write.
    // Producers cut the batch into T consecutive slices and sort their insertions by owner; an owner walks the T slices in
    // order, which is node order.
    void link_batch(uint32_t b0, uint32_t n, const uint32_t *lists) {
        const uint32_t W = M + 1;
        const uint32_t T = (uint32_t)std::max(1, threads);
        const auto t0 = timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        if (T == 1 || n < 8 * T) {                        // the definition: node by node
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t *l = lists + (size_t)i * W;
                supply[b0 + i].assign(l + 1, l + 1 + l[0]);
                if (!dirty.empty()) dirty[b0 + i] = 1;
                for (uint32_t k = 0; k < l[0]; ++k) {
                    apply_insert(supply[l[1 + k]], l[1 + k], b0 + i, 2 * M, true);
                    if (!dirty.empty()) dirty[l[1 + k]] = 1;
                }
            }
            if (timing) ns_reverse += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            return;
        }
        std::vector<std::vector<uint32_t>> bucket((size_t)T * T);   // [producer][owner]: (des, src) pairs
        run_team([&](int t, Team &team) {
            int sense = 0;
            const uint32_t lo = (uint32_t)((uint64_t)n * t / T), hi = (uint32_t)((uint64_t)n * (t + 1) / T);
            for (uint32_t i = lo; i < hi; ++i) {
                const uint32_t *l = lists + (size_t)i * W;
                for (uint32_t k = 0; k < l[0]; ++k) {
                    std::vector<uint32_t> &bk = bucket[(size_t)t * T + owner_of(l[1 + k], T)];
                    bk.push_back(l[1 + k]);
                    bk.push_back(b0 + i);
                }
            }
            team.barrier(sense);
            for (uint32_t i = 0; i < n; ++i)              // the writes of the lists this thread owns
                if (owner_of(b0 + i, T) == (uint32_t)t) {
                    const uint32_t *l = lists + (size_t)i * W;
                    supply[b0 + i].assign(l + 1, l + 1 + l[0]);
                    if (!dirty.empty()) dirty[b0 + i] = 1;
                }
            for (uint32_t p = 0; p < T; ++p) {
                const std::vector<uint32_t> &bk = bucket[(size_t)p * T + t];
                for (size_t e = 0; e + 1 < bk.size(); e += 2) {
                    // the lists are scattered over the heap: ask for the vector header 16 insertions ahead, its storage 8 ahead
                    if (e + 32 < bk.size()) __builtin_prefetch(&supply[bk[e + 32]], 0, 1);
                    if (e + 16 < bk.size()) { const std::vector<uint32_t> &nx = supply[bk[e + 16]]; if (!nx.empty()) { __builtin_prefetch(nx.data(), 1, 1); __builtin_prefetch(nx.data() + 16, 1, 1); } }
                    const uint32_t des = bk[e], src = bk[e + 1];
                    if (des >= b0 && des < b0 + n && src < des) continue;   // overwritten by des's own write
                    apply_insert(supply[des], des, src, 2 * M, true);
                    if (!dirty.empty()) dirty[des] = 1;                     // (one writer per row: its owner)
                }
            }
        });
        if (timing) ns_reverse += (long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() * (double)T);
    }

    // the batches of phase 3 (also rg_build_schedule): `batch` nodes each if the caller fixed it; otherwise the first 2,048
    // nodes one by one (the projection graph is barely connected before phase 3: searches from the entry point expand a
    // handful of nodes), then batches of at most a quarter of what is linked, between 512 and B nodes
    static std::vector<uint32_t> phase3_schedule(uint32_t nd, uint32_t batch) {
        std::vector<uint32_t> s;
        const uint32_t B = batch ? batch : std::max<uint32_t>(8192, std::min<uint32_t>(131072, nd / 24));
        const uint32_t warm = batch ? 0u : std::min<uint32_t>(nd, 2048u);
        s.assign(warm, 1u);
        for (uint32_t b0 = warm, n = 0; b0 < nd; b0 += n) {
            n = batch ? B : std::min(B, std::max<uint32_t>(512, b0 / 4));
            n = std::min(n, nd - b0);
            s.push_back(n);
        }
        return s;
    }

    // :1846-1940
    void prune_search(std::vector<Nb> &pool, uint32_t node, std::vector<uint32_t> &out) const {
        std::vector<uint32_t> result;
        if (pool.empty()) { out = result; return; }
        std::sort(pool.begin(), pool.end());
        result.reserve(2 * M);
        uint32_t start = 0;
        if (pool[start].id == node) ++start;
        const std::vector<uint32_t> &have = proj[node];
        while (start < pool.size() && has(have, pool[start].id)) ++start;
        if (start >= pool.size()) { out = result; return; }
        const uint32_t first = start;
        result.push_back(pool[start].id);
        for (uint32_t j = start + 1; j < pool.size() && j <= start + 6; ++j) prefetch_row(pool[j].id);
        while (result.size() < M && (++start) < pool.size()) {
            const Nb &p = pool[start];
            if (start + 6 < pool.size()) prefetch_row(pool[start + 6].id);
            if (!occluded(p, result) && p.id != node) result.push_back(p.id);
        }
        // second sweep (:1912-1926): only the entries in front of the first sweep's starting point are still undecided
        // (the ones skipped because the projection list holds them already); index 0 is never visited (++start first)
        start = 0;
        while (result.size() < M && (++start) < first) {
            const Nb &p = pool[start];
            if (!occluded(p, result) && p.id != node && !has(result, p.id)) result.push_back(p.id);
        }
        out = result;
    }

    // node's current list -> (id, distance to node) without repeated ids (:1112-1121, :1229-1238)
    void scored_unique(const std::vector<uint32_t> &lst, uint32_t node, std::vector<Nb> &out) const {
        out.clear();
        std::vector<uint32_t> seen;
        for (uint32_t id : lst) {
            if (has(seen, id)) continue;
            seen.push_back(id);
            out.push_back(Nb{id, cmp(id, node)});
        }
    }

    // SearchProjectionGraphInternal (:1279-1350) over the live supply graph; returns the expanded nodes in pop order
    void search_live(uint32_t node, std::vector<uint32_t> &seen, uint32_t &serial, std::vector<Nb> &expanded) {
        if (seen.empty()) seen.assign(nd, 0u);
        const uint32_t tag = ++serial;
        Beam beam;
        beam.reset(L);
        expanded.clear();
        expanded.reserve(L);
        beam.insert(ep, cmp(ep, node));
        seen[ep] = tag;
        while (beam.has_open()) {
            const Beam::E cur = beam.pop();
            expanded.push_back(Nb{cur.id, cur.dist});
            std::vector<uint32_t> nbrs;
            {   // the reference iterates supply_nbrs_[cur] unlocked; take a snapshot so a concurrent writer cannot tear it
                std::lock_guard<std::mutex> guard(locks[cur.id]);
                nbrs = supply[cur.id];
            }
            for (uint32_t nb : nbrs)
                if (seen[nb] != tag && nb != node) prefetch_row(nb);
            for (uint32_t nb : nbrs) {
                if (seen[nb] == tag || nb == node) continue;
                seen[nb] = tag;
                beam.insert(nb, cmp(nb, node));
            }
        }
    }
    // the same search over a frozen ELL snapshot (verification of the GPU batches)
    void search_snapshot(uint32_t node, const uint32_t *ell, uint32_t S, std::vector<uint32_t> &seen, uint32_t &serial,
                         std::vector<Nb> &expanded) const {
        if (seen.empty()) seen.assign(nd, 0u);
        const uint32_t tag = ++serial;
        Beam beam;
        beam.reset(L);
        expanded.clear();
        beam.insert(ep, cmp(ep, node));
        seen[ep] = tag;
        while (beam.has_open()) {
            const Beam::E cur = beam.pop();
            expanded.push_back(Nb{cur.id, cur.dist});
            const uint32_t *row = ell + (size_t)cur.id * S;
            for (uint32_t j = 0; j < row[0]; ++j) {
                const uint32_t nb = row[1 + j];
                if (seen[nb] == tag || nb == node) continue;
                seen[nb] = tag;
                beam.insert(nb, cmp(nb, node));
            }
        }
    }
    // rest of the per-node work of phase 3 (:1203-1215)
    std::atomic<long long> ns_prune{0}, ns_reverse{0};   // RG_BUILD_TIMING: thread time inside the two halves of the linking
    void link_from_search(uint32_t node, std::vector<Nb> &expanded) {
        const auto t0 = timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        expanded.erase(std::remove_if(expanded.begin(), expanded.end(), [&](const Nb &x) { return x.id == node; }), expanded.end());
        std::vector<uint32_t> pruned;
        prune_search(expanded, node, pruned);
        {
            std::lock_guard<std::mutex> guard(locks[node]);
            supply[node] = pruned;
        }
        const auto t1 = timing ? std::chrono::steady_clock::now() : t0;
        add_reverse(supply, node, 2 * M, true);
        if (timing) {
            const auto t2 = std::chrono::steady_clock::now();
            ns_prune += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
            ns_reverse += std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
        }
    }

    // Phase 3 with the n beam searches on the GPU (K1 in build mode), in batches: every node of a batch searches the
    // supply graph as it stood when the batch started (the reference's multi-threaded build sees a similarly racy
    // graph; at one thread it sees every earlier node's links -- so this variant is NOT the T=1 result, it is a valid
    // scheduling of the same algorithm).  The occlusion pruning of the expansion lists follows the searches on the GPU
    // (rg_build_prune.hip: the same lists prune_search returns, checked under RG_BUILD_VERIFY; RG_BUILD_HOST_PRUNE=1 keeps
    // it on the host threads, as do shapes the kernel does not take); reverse-edge insertion stays on the host threads.
    int gpu_device = -1;
    uint32_t gpu_batch = 0;
    bool gpu_verify = false;
    std::string gpu_error;
    bool phase3_gpu() {
        auto fail = [&](const std::string &m) { gpu_error = m; return false; };
        const uint32_t S = (2 * M + 1 + 15) / 16 * 16;
        const uint32_t cap = (2 * L + 63) / 64 * 64;
        // batch schedule: a batch never exceeds a quarter of the nodes already linked (the graph changes fastest early
        // on, when a stale snapshot hurts most), between 2,048 and Bmax nodes; a fixed size if the caller gave one
        const std::vector<uint32_t> sched = phase3_schedule(nd, gpu_batch);
        const uint32_t B = *std::max_element(sched.begin(), sched.end());
        if (hipSetDevice(gpu_device) != hipSuccess) return fail("cannot select the build GPU");
        float *d_base = nullptr;
        uint2_pod *d_exp = nullptr;
        uint32_t *d_nexp = nullptr;
        rg_index *ix = nullptr;
        std::vector<uint32_t> h_ell((size_t)nd * S);
        uint2_pod *h_exp = nullptr;     // pinned: the downloads of one half overlap the search of the other
        uint32_t *h_nexp = nullptr;
        const uint32_t hs = 72;         // row stride of the projection lists handed to the pruning kernel (length + <= 64 ids)
        uint32_t *d_have = nullptr, *h_have = nullptr, *d_out = nullptr, *h_out = nullptr;
        hipStream_t st = nullptr;
        hipEvent_t evA = nullptr, evB = nullptr;
        // the graph snapshot on the device is refreshed with the rows that changed since the last batch (`dirty`), through
        // two pinned staging buffers of CH rows (+ their row numbers) and a scatter kernel -- not with all nd rows from
        // pageable memory every batch (10M nodes: 98 batches x 3.2 GB)
        const uint32_t CH = std::min<uint32_t>(nd, 1u << 19);
        uint32_t *h_stage[2] = {nullptr, nullptr}, *d_stage[2] = {nullptr, nullptr};
        hipEvent_t evS[2] = {nullptr, nullptr};
        bool stage_used[2] = {false, false};
        std::vector<uint32_t> dl;
        bool ok = true;
        for (int i = 0; i < 2 && ok; ++i)
            ok = hipHostMalloc(&h_stage[i], (size_t)CH * (S + 1) * 4) == hipSuccess && hipMalloc(&d_stage[i], (size_t)CH * (S + 1) * 4) == hipSuccess &&
                 hipEventCreate(&evS[i]) == hipSuccess;
        dirty.assign(nd, 1);
        if (d_base_pre) { d_base = d_base_pre; d_base_pre = nullptr; }
        else ok = ok && hipMalloc(&d_base, (size_t)nd * stride * 4) == hipSuccess &&
                  rg::upload_staged(d_base, base, (size_t)nd * stride * 4) == RG_OK;      // (through pinned chunks: rg_mem.h)
        ok = ok && hipMalloc(&d_exp, (size_t)B * cap * 8) == hipSuccess && hipMalloc(&d_nexp, (size_t)B * 4) == hipSuccess &&
                  hipHostMalloc(&h_exp, (size_t)B * cap * 8) == hipSuccess && hipHostMalloc(&h_nexp, (size_t)B * 4) == hipSuccess &&
                  hipStreamCreate(&st) == hipSuccess && hipEventCreate(&evA) == hipSuccess && hipEventCreate(&evB) == hipSuccess;
        if (ok) ok = build_index_create(d_base, nd, dim, (uint32_t)stride, ep, l2 ? RG_METRIC_L2 : RG_METRIC_IP, gpu_device, S, &ix) == RG_OK;
        const bool gpu_prune = ok && !getenv("RG_BUILD_HOST_PRUNE") && build_prune_supported(ix, M, cap);
        if (gpu_prune)
            ok = hipMalloc(&d_have, (size_t)B * hs * 4) == hipSuccess && hipHostMalloc(&h_have, (size_t)B * hs * 4) == hipSuccess &&
                 hipMalloc(&d_out, (size_t)B * (M + 1) * 4) == hipSuccess && hipHostMalloc(&h_out, (size_t)B * (M + 1) * 4) == hipSuccess;
        const bool want_exp = !gpu_prune || gpu_verify;   // the expansion lists come down only if the host needs them
        std::atomic<uint32_t> mismatches(0), prune_mismatches(0);
        std::vector<std::vector<uint32_t>> stamp(std::max(1, threads));
        std::vector<uint32_t> serial(std::max(1, threads), 0);
        // The projection graph is barely connected before phase 3 (searches from the entry point expand a handful of
        // nodes), so the first nodes are linked one after another on the host, exactly as the reference does at one
        // thread; the GPU takes over once the graph is navigable, in batches of at most a quarter of what is linked.
        size_t si = 0;
        uint32_t warm = 0;
        if (!gpu_batch)
            for (; ok && si < sched.size() && sched[si] == 1; ++si, ++warm) {
                std::vector<Nb> expanded;
                search_live(warm, stamp[0], serial[0], expanded);
                link_from_search(warm, expanded);
            }
        std::vector<uint32_t> lk((size_t)B * (M + 1));   // the batch's pruned lists, GPU's or host's, as link_batch wants them
        double t_snap = 0, t_gpu = 0, t_link = 0;
        uint32_t nbatches = 0;
        size_t n_dirty = 0;
        auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
        auto t_b = std::chrono::steady_clock::now();
        if (timing) fprintf(stderr, "[rg_build]   phase 3 upload + warm-up  %8.2f s\n", since(t_mark));
        for (uint32_t b0 = warm, n = 0; ok && b0 < nd; b0 += n, ++si) {
            n = sched[si];
            t_link += since(t_b); t_b = std::chrono::steady_clock::now();
            dl.clear();
            for (uint32_t i = 0; i < nd; ++i) if (dirty[i]) { dl.push_back(i); dirty[i] = 0; }
            n_dirty += dl.size();
            for (size_t c0 = 0, ci = 0; ok && c0 < dl.size(); c0 += CH, ++ci) {
                const int bi = (int)(ci & 1);
                const uint32_t cn = (uint32_t)std::min<size_t>(CH, dl.size() - c0);
                if (stage_used[bi]) ok = hipEventSynchronize(evS[bi]) == hipSuccess;   // its last upload has left the pinned buffer
                if (!ok) break;
                uint32_t *rows = h_stage[bi], *idx = h_stage[bi] + (size_t)CH * S;
                parallel_for(cn, 4096, [&](uint32_t r, int) {
                    const uint32_t i = dl[c0 + r];
                    uint32_t *row = h_ell.data() + (size_t)i * S;
                    const std::vector<uint32_t> &l = supply[i];
                    row[0] = (uint32_t)l.size();
                    std::memcpy(row + 1, l.data(), l.size() * 4);
                    std::memcpy(rows + (size_t)r * S, row, (size_t)S * 4);
                    idx[r] = i;
                });
                ok = hipMemcpyAsync(d_stage[bi], rows, (size_t)cn * S * 4, hipMemcpyHostToDevice, st) == hipSuccess &&
                     hipMemcpyAsync(d_stage[bi] + (size_t)CH * S, idx, (size_t)cn * 4, hipMemcpyHostToDevice, st) == hipSuccess &&
                     hipEventRecord(evS[bi], st) == hipSuccess &&
                     build_index_update_rows(ix, d_stage[bi], d_stage[bi] + (size_t)CH * S, cn, st) == RG_OK;
                stage_used[bi] = true;
            }
            if (!ok) break;
            t_snap += since(t_b); t_b = std::chrono::steady_clock::now();
            // the batch goes to the GPU in two halves over the same snapshot: the host links the first half while the
            // GPU searches the second (same semantics as one launch: every node of the batch searched the snapshot)
            const uint32_t nA = n >= 4096 ? n / 2 : n, nB = n - nA;
            if (gpu_prune) {   // projection lists of the batch's nodes (constant during phase 3)
                parallel_for(n, 4096, [&](uint32_t i, int) {
                    uint32_t *row = h_have + (size_t)i * hs;
                    const std::vector<uint32_t> &l = proj[b0 + i];
                    row[0] = (uint32_t)l.size();                       // > 64: the kernel leaves the node to the host
                    std::memcpy(row + 1, l.data(), std::min<size_t>(l.size(), hs - 1) * 4);
                });
                ok = hipMemcpyAsync(d_have, h_have, (size_t)n * hs * 4, hipMemcpyHostToDevice, st) == hipSuccess;
            }
            // one half: searches, pruning, downloads, event
            auto enqueue = [&](uint32_t h0, uint32_t hn, hipEvent_t ev) {
                bool good = build_search_dev(ix, b0 + h0, hn, L, d_exp + (size_t)h0 * cap, cap, d_nexp + h0, st) == RG_OK &&
                            hipMemcpyAsync(h_nexp + h0, d_nexp + h0, (size_t)hn * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
                if (good && gpu_prune)
                    good = build_prune_dev(ix, b0 + h0, hn, M, d_exp + (size_t)h0 * cap, cap, d_nexp + h0, d_have + (size_t)h0 * hs, hs,
                                           d_out + (size_t)h0 * (M + 1), st) == RG_OK &&
                           hipMemcpyAsync(h_out + (size_t)h0 * (M + 1), d_out + (size_t)h0 * (M + 1), (size_t)hn * (M + 1) * 4,
                                          hipMemcpyDeviceToHost, st) == hipSuccess;
                if (good && want_exp)
                    good = hipMemcpyAsync(h_exp + (size_t)h0 * cap, d_exp + (size_t)h0 * cap, (size_t)hn * cap * 8, hipMemcpyDeviceToHost, st) == hipSuccess;
                return good && hipEventRecord(ev, st) == hipSuccess;
            };
            ok = ok && enqueue(0, nA, evA);
            if (ok && nB) ok = enqueue(nA, nB, evB);
            if (ok) ok = hipEventSynchronize(evA) == hipSuccess;
            if (!ok) break;
            t_gpu += since(t_b); t_b = std::chrono::steady_clock::now();
            ++nbatches;
            for (int half = 0; half < (nB ? 2 : 1) && ok; ++half) {
            const uint32_t h0 = half ? nA : 0u, hn = half ? nB : nA;
            if (half) {
                t_link += since(t_b); t_b = std::chrono::steady_clock::now();
                ok = hipEventSynchronize(evB) == hipSuccess;
                t_gpu += since(t_b); t_b = std::chrono::steady_clock::now();
                if (!ok) break;
            }
            parallel_for(hn, 64, [&](uint32_t ii, int t) {
                const uint32_t i = h0 + ii;
                const uint32_t node = b0 + i;
                std::vector<Nb> expanded;
                const uint32_t *po = gpu_prune ? h_out + (size_t)i * (M + 1) : nullptr;
                uint32_t *dst = lk.data() + (size_t)i * (M + 1);
                if (po && po[0] != 0xffffffffu && !gpu_verify) {
                    std::memcpy(dst, po, (size_t)(po[0] + 1) * 4);
                    return;
                }
                if (h_nexp[i] > cap || (po && po[0] == 0xffffffffu && !want_exp)) {
                    search_snapshot(node, h_ell.data(), S, stamp[t], serial[t], expanded);   // list did not fit / left to the host
                } else {
                    expanded.resize(h_nexp[i]);
                    const uint2_pod *e = h_exp + (size_t)i * cap;
                    for (uint32_t j = 0; j < h_nexp[i]; ++j) {
                        float d;
                        std::memcpy(&d, &e[j].x, 4);
                        expanded[j] = Nb{e[j].y, d};
                    }
                    if (gpu_verify) {
                        std::vector<Nb> ref;
                        search_snapshot(node, h_ell.data(), S, stamp[t], serial[t], ref);
                        bool same = ref.size() == expanded.size();
                        for (size_t j = 0; same && j < ref.size(); ++j)
                            same = ref[j].id == expanded[j].id && std::memcmp(&ref[j].dist, &expanded[j].dist, 4) == 0;
                        if (!same) mismatches.fetch_add(1);
                        if (po && po[0] != 0xffffffffu) {   // the GPU's pruned list against prune_search on the same expansion list
                            std::vector<Nb> pool = expanded;
                            pool.erase(std::remove_if(pool.begin(), pool.end(), [&](const Nb &x) { return x.id == node; }), pool.end());
                            std::vector<uint32_t> want;
                            prune_search(pool, node, want);
                            if (want.size() != po[0] || !std::equal(want.begin(), want.end(), po + 1)) prune_mismatches.fetch_add(1);
                        }
                    }
                }
                // the host's pruning of this node's expansion list (shapes or nodes the kernel left to the host, or the check)
                expanded.erase(std::remove_if(expanded.begin(), expanded.end(), [&](const Nb &x) { return x.id == node; }), expanded.end());
                std::vector<uint32_t> pruned;
                prune_search(expanded, node, pruned);
                dst[0] = (uint32_t)pruned.size();
                std::memcpy(dst + 1, pruned.data(), pruned.size() * 4);
            });
            link_batch(b0 + h0, hn, lk.data() + (size_t)h0 * (M + 1));
            }
        }
        t_link += since(t_b);
        if (timing) fprintf(stderr, "[rg_build]   phase 3: %u batches, snapshot (%.1f M changed rows staged + uploaded) %.2f s, GPU search%s + download %.2f s, host linking %.2f s\n",
                            nbatches, n_dirty * 1e-6, t_snap, gpu_prune ? " + GPU pruning" : "", t_gpu, t_link);
        dirty.clear(); dirty.shrink_to_fit();
        if (st) (void)hipStreamSynchronize(st);
        for (int i = 0; i < 2; ++i) {
            if (h_stage[i]) (void)hipHostFree(h_stage[i]);
            if (d_stage[i]) (void)hipFree(d_stage[i]);
            if (evS[i]) (void)hipEventDestroy(evS[i]);
        }
        if (ix) rg_index_close(ix);
        if (d_base) (void)hipFree(d_base);
        if (d_exp) (void)hipFree(d_exp);
        if (d_nexp) (void)hipFree(d_nexp);
        if (st) (void)hipStreamSynchronize(st);
        if (h_exp) (void)hipHostFree(h_exp);
        if (h_nexp) (void)hipHostFree(h_nexp);
        if (d_have) (void)hipFree(d_have);
        if (d_out) (void)hipFree(d_out);
        if (h_have) (void)hipHostFree(h_have);
        if (h_out) (void)hipHostFree(h_out);
        if (evA) (void)hipEventDestroy(evA);
        if (evB) (void)hipEventDestroy(evB);
        if (st) (void)hipStreamDestroy(st);
        if (!ok) return fail(std::string("GPU phase 3 failed: ") + rg_last_error());
        if (mismatches.load()) return fail("GPU phase 3: " + std::to_string(mismatches.load()) + " expansion lists differ from the host search");
        if (prune_mismatches.load()) return fail("GPU phase 3: " + std::to_string(prune_mismatches.load()) + " pruned lists differ from the host pruning");
        return true;
    }

    // Phase 3 on host threads, many of them: the batches of phase3_schedule, every node of a batch searching the graph as it
    // stood when the batch began (a frozen flat copy), then link_batch -- what phase3_gpu does with the searches and the
    // pruning on the host, so the two builders (and the oracle given the same schedule) write the same index.
    void phase3_host_batched() {
        const std::vector<uint32_t> sched = phase3_schedule(nd, 0);
        const uint32_t S = 2 * M + 1;
        std::vector<std::vector<uint32_t>> stamp(std::max(1, threads));
        std::vector<uint32_t> serial(std::max(1, threads), 0);
        size_t si = 0;
        uint32_t b0 = 0;
        for (; si < sched.size() && sched[si] == 1; ++si, ++b0) {
            std::vector<Nb> expanded;
            search_live(b0, stamp[0], serial[0], expanded);
            link_from_search(b0, expanded);
        }
        if (b0 >= nd) return;
        std::vector<uint32_t> ell((size_t)nd * S), lk;
        for (; si < sched.size(); b0 += sched[si], ++si) {
            const uint32_t n = sched[si];
            parallel_for(nd, 4096, [&](uint32_t i, int) {
                uint32_t *row = ell.data() + (size_t)i * S;
                const std::vector<uint32_t> &l = supply[i];
                row[0] = (uint32_t)l.size();
                std::memcpy(row + 1, l.data(), l.size() * 4);
            });
            lk.resize((size_t)n * (M + 1));
            parallel_for(n, 16, [&](uint32_t i, int t) {
                const uint32_t node = b0 + i;
                std::vector<Nb> expanded;
                search_snapshot(node, ell.data(), S, stamp[t], serial[t], expanded);
                expanded.erase(std::remove_if(expanded.begin(), expanded.end(), [&](const Nb &x) { return x.id == node; }), expanded.end());
                std::vector<uint32_t> pruned;
                prune_search(expanded, node, pruned);
                uint32_t *dst = lk.data() + (size_t)i * (M + 1);
                dst[0] = (uint32_t)pruned.size();
                std::memcpy(dst + 1, pruned.data(), pruned.size() * 4);
            });
            link_batch(b0, n, lk.data());
        }
    }

    // RG_BUILD_TIMING=1: per-phase wall times on stderr
    bool timing = getenv("RG_BUILD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t_mark = std::chrono::steady_clock::now();
    void lap(const char *what) {
        const auto now = std::chrono::steady_clock::now();
        if (timing) fprintf(stderr, "[rg_build] %-28s %8.2f s\n", what, std::chrono::duration<double>(now - t_mark).count());
        t_mark = now;
    }

    bool run(const uint32_t *knn, uint32_t nq, uint32_t kdim) {
        lap("setup");
        proj.assign(nd, {});
        supply.assign(nd, {});
        locks = std::vector<std::mutex>(nd);
        // ---- entry point (:2004-2041): float sums in index order, plain (unfused) arithmetic -- on the build GPU when there
        // is one (rg_projection_ep_dev: the same sums in the same order, checked bit for bit against this loop in the tests)
        if (gpu_device >= 0 && dim % 4 == 0 && stride % 4 == 0 && hipSetDevice(gpu_device) == hipSuccess &&
            hipMalloc(&d_base_pre, (size_t)nd * stride * 4) == hipSuccess) {
            if (rg::upload_staged(d_base_pre, base, (size_t)nd * stride * 4) == RG_OK &&
                rg_projection_ep_dev(d_base_pre, nd, dim, (uint32_t)stride, gpu_device, &ep) == RG_OK)
                ep_known = true;
            else { (void)hipFree(d_base_pre); d_base_pre = nullptr; }
        }
        if (!ep_known) rg_projection_ep(base, nd, dim, (uint32_t)stride, &ep);
        lap("entry point");
        // ---- phase 1 (:1059-1097): every training query links its nearest base point to its other neighbours.
        // GPU-assisted build (round 3): the pruned list of a query depends on the base and on its knn row only, not on the
        // graph, so all of them are computed up front on the GPU (rg_build_prune.hip: rg_knn_score_kernel + the pruning
        // kernel in its PruneBiSearchBaseGetBase form; the lists prune_get_base returns, checked under RG_BUILD_VERIFY);
        // what stays on the host is the part that does depend on the graph: the assignment and the reverse edges.
        std::vector<uint32_t> p1;          // [nq][M + 1]: length (0xffffffff: this query is pruned on the host) + ids
        std::atomic<uint32_t> p1_mismatches(0);
        if (gpu_device >= 0 && d_base_pre && !getenv("RG_BUILD_HOST_PRUNE") && nq > 0) {
            const uint32_t ncol = std::min(kdim, Nq);
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, gpu_device) == hipSuccess && build_prune_knn_supported(dim, M, ncol, 160 * 1024)) {
                const uint32_t chunk = 262144;
                uint32_t *d_knn = nullptr, *d_piv = nullptr, *d_out = nullptr;
                uint2_pod *d_exp = nullptr;
                bool ok = hipMalloc(&d_knn, (size_t)chunk * kdim * 4) == hipSuccess && hipMalloc(&d_piv, (size_t)chunk * 4) == hipSuccess &&
                          hipMalloc(&d_out, (size_t)chunk * (M + 1) * 4) == hipSuccess && hipMalloc(&d_exp, (size_t)chunk * ncol * 8) == hipSuccess;
                if (ok) p1.assign((size_t)nq * (M + 1), 0xffffffffu);
                for (uint32_t q0 = 0; ok && q0 < nq; q0 += chunk) {
                    const uint32_t n = std::min(chunk, nq - q0);
                    ok = hipMemcpy(d_knn, knn + (size_t)q0 * kdim, (size_t)n * kdim * 4, hipMemcpyHostToDevice) == hipSuccess &&
                         build_prune_knn_dev(d_base_pre, dim, (uint32_t)stride, l2 ? RG_METRIC_L2 : RG_METRIC_IP, gpu_device, prop.multiProcessorCount,
                                             160 * 1024, d_knn, n, kdim, ncol, M, d_exp, ncol, d_piv, d_out, nullptr) == RG_OK &&
                         hipMemcpy(p1.data() + (size_t)q0 * (M + 1), d_out, (size_t)n * (M + 1) * 4, hipMemcpyDeviceToHost) == hipSuccess;
                }
                for (void *ptr : {(void *)d_knn, (void *)d_piv, (void *)d_out, (void *)d_exp}) if (ptr) (void)hipFree(ptr);
                if (!ok) { (void)hipGetLastError(); p1.clear(); }      // the host prunes everything, as before
            }
            lap("phase 1 pruning (GPU)");
        }
        auto phase1_query = [&](uint32_t sq) {
            const uint32_t n = std::min(kdim, Nq);
            if (n == 0) return;
            const uint32_t *nn = knn + (size_t)sq * kdim;
            const uint32_t tgt = nn[0];
            std::vector<uint32_t> pruned;
            const uint32_t *po = p1.empty() ? nullptr : p1.data() + (size_t)sq * (M + 1);
            if (po && po[0] != 0xffffffffu) pruned.assign(po + 1, po + 1 + po[0]);
            if (!po || po[0] == 0xffffffffu || gpu_verify) {
                for (uint32_t i = 0; i < n; ++i) prefetch_row(nn[i]);
                std::vector<Nb> full;
                for (uint32_t i = 0; i < n; ++i)
                    if (nn[i] != tgt) full.push_back(Nb{nn[i], cmp(nn[i], tgt)});
                std::vector<uint32_t> host_list;
                prune_get_base(full, tgt, host_list);
                if (po && po[0] != 0xffffffffu && host_list != pruned) p1_mismatches.fetch_add(1);
                pruned.swap(host_list);
            }
            {
                std::lock_guard<std::mutex> guard(locks[tgt]);
                proj[tgt] = pruned;
            }
            add_reverse(proj, tgt, M, false);
        };
        if (threads <= 1) {
            for (uint32_t sq = 0; sq < nq; ++sq) phase1_query(sq);   // the reference's order
        } else if (std::min(kdim, Nq) > 0) {
            // the one-thread result on many threads (phase1_replay): first every query's pruned list (the GPU's, or the host's
            // for the queries the kernel left out / for the check), then each base point's list replayed on its own
            const uint32_t n = std::min(kdim, Nq);
            if (p1.empty()) p1.assign((size_t)nq * (M + 1), 0xffffffffu);
            parallel_for(nq, 256, [&](uint32_t sq, int) {
                uint32_t *po = p1.data() + (size_t)sq * (M + 1);
                if (po[0] != 0xffffffffu && !gpu_verify) return;
                const uint32_t *nn = knn + (size_t)sq * kdim;
                const uint32_t tgt = nn[0];
                for (uint32_t i = 0; i < n; ++i) prefetch_row(nn[i]);
                std::vector<Nb> full;
                for (uint32_t i = 0; i < n; ++i)
                    if (nn[i] != tgt) full.push_back(Nb{nn[i], cmp(nn[i], tgt)});
                std::vector<uint32_t> host_list;
                prune_get_base(full, tgt, host_list);
                if (po[0] != 0xffffffffu && (po[0] != host_list.size() || !std::equal(host_list.begin(), host_list.end(), po + 1)))
                    p1_mismatches.fetch_add(1);
                po[0] = (uint32_t)host_list.size();
                std::memcpy(po + 1, host_list.data(), host_list.size() * 4);
            });
            lap("phase 1 pruning (host part)");
            phase1_replay(knn, nq, kdim, p1);
        }
        lap("phase 1");
        if (p1_mismatches.load()) { gpu_error = "GPU phase 1: " + std::to_string(p1_mismatches.load()) + " pruned lists differ from the host pruning"; return false; }
        p1.clear(); p1.shrink_to_fit();
        // ---- phase 2 (:1100-1136)
        if (threads <= 1) for (uint32_t node = 0; node < nd; ++node) add_reverse(proj, node, M, false);
        else phase2_windows();
        lap("phase 2 reverse edges");
        parallel_for(nd, 2048, [&](uint32_t node, int) {
            if (proj[node].size() <= M) return;
            std::vector<Nb> full;
            scored_unique(proj[node], node, full);
            full.erase(std::remove_if(full.begin(), full.end(), [&](const Nb &x) { return x.id == node; }), full.end());
            std::vector<uint32_t> pruned;
            prune_get_base(full, node, pruned);
            std::lock_guard<std::mutex> guard(locks[node]);
            proj[node] = pruned;
        });
        lap("phase 2 pruning");
        parallel_for(nd, 8192, [&](uint32_t i, int) { supply[i] = proj[i]; });   // :1183-1188
        lap("supply copy");
        // ---- phase 3 (:1192-1220): connectivity enhancement -- beam search from the entry point towards every node
        if (gpu_device >= 0) {
            if (!phase3_gpu()) return false;
        } else if (threads <= 1) {
            std::vector<uint32_t> stamp;
            uint32_t serial = 0;
            for (uint32_t node = 0; node < nd; ++node) {     // the reference's one-thread sequence
                std::vector<Nb> expanded;
                search_live(node, stamp, serial, expanded);
                link_from_search(node, expanded);
            }
        } else {
            phase3_host_batched();
        }
        lap("phase 3");
        if (timing) fprintf(stderr, "[rg_build]   linking, thread-seconds: prune of the expansion list %.1f, reverse edges %.1f\n",
                            ns_prune.load() * 1e-9, ns_reverse.load() * 1e-9);
        // ---- phase 4 (:1224-1248)
        parallel_for(nd, 2048, [&](uint32_t node, int) {
            if (supply[node].size() <= M) return;
            std::vector<Nb> full;
            scored_unique(supply[node], node, full);
            std::vector<uint32_t> pruned;
            prune_search(full, node, pruned);
            std::lock_guard<std::mutex> guard(locks[node]);
            supply[node] = pruned;
        });
        lap("phase 4");
        // ---- phase 5 (:1251-1264): append up to 2*M supply edges that the projection list lacks
        parallel_for(nd, 100, [&](uint32_t i, int) {
            std::vector<uint32_t> ok;
            for (uint32_t s : supply[i]) {
                if (ok.size() >= 2 * M) break;
                if (!has(proj[i], s)) ok.push_back(s);
            }
            proj[i].insert(proj[i].end(), ok.begin(), ok.end());
        });
        lap("phase 5");
        return true;
    }
};

}  // namespace
}  // namespace rg

// CPUs this process can actually run on at once: its affinity mask, cut by the cgroup CPU quota when there is one (a
// container that shows 256 CPUs may be allowed 16 CPUs' worth of time; threads beyond that only take turns -- and turn
// every barrier of the deterministic phases into a wait for the scheduler)
static unsigned usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                   // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        long long period = 0;
        if (fscanf(f, "%31s %lld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) {
            const long long quota = atoll(q);
            if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        }
        fclose(f);
    } else {
        long long quota = -1, period = 0;                                    // cgroup v1
        if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
        if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
        if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    }
    return n;
}

static rg_status build_body(const float *base, uint32_t nb, uint32_t dim, uint32_t stride, const uint32_t *knn_ids,
                            uint32_t nq, uint32_t knn_k, int metric, uint32_t M_sq, uint32_t M_pjbp, uint32_t L_pjpq,
                            uint32_t num_threads, int device, uint32_t batch, uint32_t *out_ep, uint64_t **out_offsets,
                            uint32_t **out_nbrs);
// nothing C++ leaves through the C ABI: the host side allocates O(nb) vectors and starts threads, both can fail
static rg_status build_impl(const float *base, uint32_t nb, uint32_t dim, uint32_t stride, const uint32_t *knn_ids,
                            uint32_t nq, uint32_t knn_k, int metric, uint32_t M_sq, uint32_t M_pjbp, uint32_t L_pjpq,
                            uint32_t num_threads, int device, uint32_t batch, uint32_t *out_ep, uint64_t **out_offsets,
                            uint32_t **out_nbrs) {
    try {
        return build_body(base, nb, dim, stride, knn_ids, nq, knn_k, metric, M_sq, M_pjbp, L_pjpq, num_threads, device, batch, out_ep,
                          out_offsets, out_nbrs);
    } catch (const std::bad_alloc &) {
        return rg::set_error(RG_ERR_OOM, "out of host memory during the build");
    } catch (const std::exception &e) {
        return rg::set_error(RG_ERR_DEVICE, std::string("build failed: ") + e.what());
    }
}
static rg_status build_body(const float *base, uint32_t nb, uint32_t dim, uint32_t stride, const uint32_t *knn_ids,
                            uint32_t nq, uint32_t knn_k, int metric, uint32_t M_sq, uint32_t M_pjbp, uint32_t L_pjpq,
                            uint32_t num_threads, int device, uint32_t batch, uint32_t *out_ep, uint64_t **out_offsets,
                            uint32_t **out_nbrs) {
    using rg::set_error;
    if (!base || !knn_ids || !out_ep || !out_offsets || !out_nbrs) return set_error(RG_ERR_ARG, "null argument");
    if (nb == 0 || dim == 0 || stride < dim || M_pjbp == 0 || L_pjpq == 0 || knn_k == 0)
        return set_error(RG_ERR_ARG, "bad build parameters");
    if (metric != RG_METRIC_L2 && metric != RG_METRIC_IP && metric != RG_METRIC_COSINE)
        return set_error(RG_ERR_ARG, "Unknown distance type");
    if (device >= 0 && (dim % 8 || stride % 4))
        return set_error(RG_ERR_ARG, "GPU-assisted build needs dim % 8 == 0 and stride % 4 == 0 (load the base with rg_fbin_load)");
    for (size_t i = 0; i < (size_t)nq * knn_k; ++i)
        if (knn_ids[i] >= nb) return set_error(RG_ERR_FORMAT, "learn base knn file references a base id >= npts");
    std::vector<float> normed;
    if (metric == RG_METRIC_COSINE) {   // BuildRoarGraph normalises the base in place (:175-181)
        normed.assign(base, base + (size_t)nb * stride);
        rg_normalize_rows(normed.data(), nb, stride, dim);
        base = normed.data();
    }
    rg::Builder b;
    b.base = base; b.stride = stride; b.dim = dim; b.nd = nb;
    b.l2 = metric == RG_METRIC_L2;
#if defined(__x86_64__)
    b.avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma");
#else
    b.avx512 = false;
#endif
    b.M = M_pjbp; b.L = L_pjpq; b.Nq = M_sq;
    // more than one thread: as many as asked for, but no more than can run at once (the result does not depend on the count)
    b.threads = (int)std::max<uint32_t>(1, num_threads);
    if (b.threads > 1 && !getenv("RG_BUILD_THREADS_AS_GIVEN")) b.threads = (int)std::max(2u, std::min<unsigned>((unsigned)b.threads, usable_cpus()));
    b.gpu_device = device;
    b.gpu_batch = batch;
    b.gpu_verify = getenv("RG_BUILD_VERIFY") != nullptr;
    if (!b.run(knn_ids, nq, knn_k)) return set_error(RG_ERR_DEVICE, b.gpu_error);
    size_t edges = 0;
    for (auto &l : b.proj) edges += l.size();
    uint64_t *off = (uint64_t *)std::malloc(((size_t)nb + 1) * 8);
    uint32_t *nbr = (uint32_t *)std::malloc(std::max<size_t>(edges * 4, 4));
    if (!off || !nbr) { std::free(off); std::free(nbr); return set_error(RG_ERR_OOM, "out of host memory"); }
    size_t pos = 0;
    for (uint32_t i = 0; i < nb; ++i) {
        off[i] = pos;
        std::memcpy(nbr + pos, b.proj[i].data(), b.proj[i].size() * 4);
        pos += b.proj[i].size();
    }
    off[nb] = pos;
    *out_ep = b.ep;
    *out_offsets = off;
    *out_nbrs = nbr;
    return RG_OK;
}

/* CalculateProjectionep (src/index_bipartite.cpp:2004-2041), host form: float sums in index order, plain (unfused)
 * arithmetic, first of equal distances */
extern "C" rg_status rg_projection_ep(const float *base, uint32_t nd, uint32_t dim, uint32_t stride, uint32_t *out_ep) {
    if (!base || !out_ep || nd == 0 || stride < dim) return rg::set_error(RG_ERR_ARG, "bad argument");
    std::vector<float> center(dim, 0.0f);
    for (size_t i = 0; i < nd; ++i)
        for (unsigned d = 0; d < dim; ++d) center[d] += base[i * stride + d];
    for (unsigned d = 0; d < dim; ++d) center[d] /= (float)nd;
    uint32_t best = 0;
    float bestd = 0.0f;
    for (size_t i = 0; i < nd; ++i) {
        float diff = 0.0f;
        for (unsigned j = 0; j < dim; ++j) {
            const float t = center[j] - base[i * stride + j];
            diff += t * t;
        }
        if (i == 0 || diff < bestd) { best = (uint32_t)i; bestd = diff; }
    }
    *out_ep = best;
    return RG_OK;
}

extern "C" rg_status rg_build_roargraph(const float *base, uint32_t nb, uint32_t dim, uint32_t stride,
                                        const uint32_t *knn_ids, uint32_t nq, uint32_t knn_k, int metric, uint32_t M_sq,
                                        uint32_t M_pjbp, uint32_t L_pjpq, uint32_t num_threads, uint32_t *out_ep,
                                        uint64_t **out_offsets, uint32_t **out_nbrs) {
    return build_impl(base, nb, dim, stride, knn_ids, nq, knn_k, metric, M_sq, M_pjbp, L_pjpq, num_threads, -1, 0, out_ep,
                      out_offsets, out_nbrs);
}

extern "C" rg_status rg_build_roargraph_gpu(const float *base, uint32_t nb, uint32_t dim, uint32_t stride,
                                            const uint32_t *knn_ids, uint32_t nq, uint32_t knn_k, int metric,
                                            uint32_t M_sq, uint32_t M_pjbp, uint32_t L_pjpq, uint32_t num_threads,
                                            int device, uint32_t batch, uint32_t *out_ep, uint64_t **out_offsets,
                                            uint32_t **out_nbrs) {
    int ndev = 0;
    if (device < 0 || hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev)
        return rg::set_error(RG_ERR_DEVICE, "no HIP device visible: the gfx950 path cannot run (there is no CPU fallback)");
    return build_impl(base, nb, dim, stride, knn_ids, nq, knn_k, metric, M_sq, M_pjbp, L_pjpq, num_threads, device, batch,
                      out_ep, out_offsets, out_nbrs);
}

extern "C" rg_status rg_build_schedule(uint32_t nb, uint32_t batch, uint32_t *sizes, uint32_t cap, uint32_t *count) {
    if (!count || (cap && !sizes)) return rg::set_error(RG_ERR_ARG, "null argument");
    const std::vector<uint32_t> s = rg::Builder::phase3_schedule(nb, batch);
    for (size_t i = 0; i < s.size() && i < cap; ++i) sizes[i] = s[i];
    *count = (uint32_t)s.size();
    return RG_OK;
}

/* One call of one pruning rule of the construction (tests: the product's rules against tests/golden/prune_*.npz, which `rg_ref prune`
 * made with the reference's own Distance / Neighbor objects).  kind 0 = PruneBiSearchBaseGetBase (:1612-1694), 1 =
 * PruneProjectionReverseCandidates (:1526-1610), 2 = PruneProjectionInternalReverseCandidates (:1434-1524), 3 =
 * PruneProjectionBaseSearchCandidates (:1846-1940; `have` = projection_graph_[pivot]).  use_gpu = 0: the host routines of the builder
 * (no GPU needed); 1: the pruning kernel (rg_build_prune.hip) as the GPU-assisted build launches it -- kind 3 from an expansion list,
 * kind 0 from a knn row whose first entry is the pivot (the distances are then computed on the device and `dists` is ignored). */
extern "C" rg_status rg_build_prune_debug(const float *base, uint32_t nb, uint32_t dim, uint32_t stride, int metric, uint32_t M, int kind, uint32_t pivot,
                                          const uint32_t *ids, const float *dists, uint32_t np, const uint32_t *have, uint32_t nhave, uint32_t *out,
                                          uint32_t *nout, int use_gpu, int device) {
    using rg::set_error;
    if (!base || !ids || !out || !nout || (np && kind != 1 && kind != 2 && !dists) || (nhave && !have)) return set_error(RG_ERR_ARG, "null argument");
    if (kind < 0 || kind > 3 || pivot >= nb || M == 0 || stride < dim) return set_error(RG_ERR_ARG, "bad argument");
    if (metric != RG_METRIC_L2 && metric != RG_METRIC_IP) return set_error(RG_ERR_ARG, "Unknown distance type");
    for (uint32_t i = 0; i < np; ++i) if (ids[i] >= nb) return set_error(RG_ERR_ARG, "id >= npts");
    try {
        if (!use_gpu) {
            rg::Builder b;
            b.base = base; b.stride = stride; b.dim = dim; b.nd = nb;
            b.l2 = metric == RG_METRIC_L2;
#if defined(__x86_64__)
            b.avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma");
#else
            b.avx512 = false;
#endif
            b.M = M;
            std::vector<uint32_t> res;
            if (kind == 1 || kind == 2) {
                res.assign(ids, ids + np);
                b.prune_reverse(pivot, res, kind == 2);
            } else {
                std::vector<rg::Nb> pool(np);
                for (uint32_t i = 0; i < np; ++i) pool[i] = rg::Nb{ids[i], dists[i]};
                if (kind == 0) b.prune_get_base(pool, pivot, res);
                else {
                    b.proj.resize((size_t)pivot + 1);
                    b.proj[pivot].assign(have, have + nhave);
                    b.prune_search(pool, pivot, res);
                }
            }
            for (size_t i = 0; i < res.size(); ++i) out[i] = res[i];
            *nout = (uint32_t)res.size();
            return RG_OK;
        }
        if (kind != 0 && kind != 3) return set_error(RG_ERR_ARG, "the GPU prunes kinds 0 and 3 only");
        if (dim % 8 || stride % 4) return set_error(RG_ERR_ARG, "GPU pruning needs dim % 8 == 0 and stride % 4 == 0");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_error(RG_ERR_DEVICE, "no HIP device visible: the gfx950 path cannot run (there is no CPU fallback)");
        if (hipSetDevice(device) != hipSuccess) return set_error(RG_ERR_DEVICE, "cannot select the device");
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return set_error(RG_ERR_DEVICE, "hipGetDeviceProperties");
        const uint32_t cap = std::max<uint32_t>(np, 1);
        struct Dev { void *p = nullptr; ~Dev() { if (p) (void)hipFree(p); } };
        Dev d_base, d_exp, d_nexp, d_have, d_out, d_knn, d_piv;
        bool ok = hipMalloc(&d_base.p, (size_t)nb * stride * 4) == hipSuccess && hipMalloc(&d_exp.p, (size_t)cap * 8) == hipSuccess &&
                  hipMalloc(&d_nexp.p, 4) == hipSuccess && hipMalloc(&d_have.p, ((size_t)nhave + 1) * 4) == hipSuccess &&
                  hipMalloc(&d_out.p, ((size_t)M + 1) * 4) == hipSuccess && hipMalloc(&d_knn.p, (size_t)cap * 4) == hipSuccess && hipMalloc(&d_piv.p, 4) == hipSuccess;
        if (!ok) return set_error(RG_ERR_OOM, "no device memory for the pruning test");
        ok = hipMemcpy(d_base.p, base, (size_t)nb * stride * 4, hipMemcpyHostToDevice) == hipSuccess;
        std::vector<uint32_t> h_out((size_t)M + 1, 0);
        rg_status st = RG_OK;
        if (kind == 3) {
            std::vector<rg::uint2_pod> ex(cap);
            for (uint32_t i = 0; i < np; ++i) { uint32_t bits; std::memcpy(&bits, &dists[i], 4); ex[i] = rg::uint2_pod{bits, ids[i]}; }
            std::vector<uint32_t> hv((size_t)nhave + 1);
            hv[0] = nhave;
            for (uint32_t i = 0; i < nhave; ++i) hv[1 + i] = have[i];
            ok = ok && hipMemcpy(d_exp.p, ex.data(), (size_t)cap * 8, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(d_nexp.p, &np, 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_have.p, hv.data(), hv.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
            if (!ok) return set_error(RG_ERR_DEVICE, "upload failed");
            rg_index ix;
            ix.d_base = (float *)d_base.p; ix.nd = nb; ix.dim = dim; ix.stride = stride; ix.metric = metric; ix.device = device;
            ix.num_cu = prop.multiProcessorCount;
            if (!rg::build_prune_supported(&ix, M, cap)) return set_error(RG_ERR_ARG, "pruning kernel: shape not supported");
            st = rg::build_prune_dev(&ix, pivot, 1, M, (const rg::uint2_pod *)d_exp.p, cap, (const uint32_t *)d_nexp.p, (const uint32_t *)d_have.p, nhave + 1,
                                     (uint32_t *)d_out.p, nullptr);
        } else {
            if (np == 0 || ids[0] != pivot) return set_error(RG_ERR_ARG, "GPU kind 0: the pool is a knn row whose first entry is the pivot");
            ok = ok && hipMemcpy(d_knn.p, ids, (size_t)np * 4, hipMemcpyHostToDevice) == hipSuccess;
            if (!ok) return set_error(RG_ERR_DEVICE, "upload failed");
            if (!rg::build_prune_knn_supported(dim, M, np, 160 * 1024)) return set_error(RG_ERR_ARG, "pruning kernel: shape not supported");
            st = rg::build_prune_knn_dev((const float *)d_base.p, dim, stride, metric, device, prop.multiProcessorCount, 160 * 1024, (const uint32_t *)d_knn.p, 1, np, np, M,
                                         (rg::uint2_pod *)d_exp.p, cap, (uint32_t *)d_piv.p, (uint32_t *)d_out.p, nullptr);
        }
        if (st != RG_OK) return st;
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h_out.data(), d_out.p, h_out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return set_error(RG_ERR_DEVICE, "the pruning kernel failed");
        if (h_out[0] == 0xffffffffu) return set_error(RG_ERR_ARG, "the pruning kernel left this list to the host (a repeated id, or a list beyond its capacity)");
        *nout = h_out[0];
        for (uint32_t i = 0; i < h_out[0]; ++i) out[i] = h_out[1 + i];
        return RG_OK;
    } catch (const std::bad_alloc &) {
        return set_error(RG_ERR_OOM, "out of host memory");
    }
}
