// rg_search.hip -- gfx950 kernels and C-ABI entry points of the search path.
//
//   K1  rg_search_kernel   persistent beam search, one wave64 per in-flight query
//                          == IndexBipartite::SearchRoarGraph (src/index_bipartite.cpp:2311-2420)
//   K1b rg_score_kernel    batched Distance::compare (include/efanna2e/distance.h:18)
//
// Per query (one wave, one single-wave workgroup, all state wave-private):
//   LDS   : sorted beam of L_pq (dist, id|expanded) pairs  == NeighborPriorityQueue (neighbor.h:138-223)
//           query vector, 64-entry candidate id/score scratch, LDS-DMA staging for the row gather
//   HBM   : epoch-tagged visited words per slot             == VisitedList (visited_list_pool.h:8-29): like the
//           reference's curV tag array, a word is "set for this query" only if its 16-bit epoch equals the query's,
//           so nothing is ever cleared between queries (a wipe happens once per 65,535 queries of a slot)
// Per hop (hop-synchronous, see SURVEY.md Appendix C-11 for why this reproduces the sequential inserts):
//   pop closest unexpanded -> read its adjacency row -> atomicOr visited bits -> ballot-compact the unvisited ids
//   -> gather + score them 4 rows per sub-pass -> rank-merge the survivors into the beam.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rg.h"
#include "rg_device.h"
#include "rg_internal.h"

namespace rg {

#define RG_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return set_error(RG_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

// device allocation released on every exit path of a host wrapper
template <typename T>
struct DevBuf {
    T *p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16)); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    T *release() { T *r = p; p = nullptr; return r; }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// ------------------------------------------------------------------------------------------------ device
struct SearchParams {
    const float *base;
    uint32_t stride, dim, nd;
    const uint32_t *ell;      // [nd][ell_stride]: word 0 = degree, then neighbour ids (null -> CSR)
    uint32_t ell_stride;
    const uint64_t *offsets;  // CSR
    const uint32_t *nbrs;
    uint32_t ep;
    const float *queries;
    uint32_t nq, qstride, k, L;
    uint32_t *out_ids;
    float *out_dists;
    uint32_t *out_cmps, *out_hops;
    uint32_t *visited;        // [slots][vwords]; word = epoch16 << 16 | 16 visited bits (nodes 16w .. 16w+15)
    uint32_t vwords;
    uint32_t *slot_epoch;     // [slots] last epoch used by the slot (persists across launches)
    uint32_t *counter;        // work-queue head
    unsigned long long *status;  // min over failing queries of (query << 32 | queue size); ~0 = none
    uint32_t stage_floats;    // floats per sub-pass staging buffer (ceil(dim/64)*256; fast mode: 256 per 128 bf16 elements)
    uint32_t stage_total;     // floats of the whole staging region (>= R * stage_floats; fast mode: >= one fp32 pass too)
    uint32_t qbase;           // index of queries[0] in the caller's batch (error reporting of chunked launches)
    uint32_t diag;            // diagnostics only (breaks parity): bit0 = skip the visited test
    uint32_t vf_slots_log2;   // log2 of the LDS visited-filter size (16-bit entries)
    uint32_t vf_front;        // VIS=0: 1 = the LDS filter screens the exact HBM words
    uint32_t *qlog;           // VIS=1, optional: [nq][logcap] ids scored by each query (input of the exact distinct count)
    uint32_t logcap;
    uint32_t *qlog_n;         // [nq] number of ids scored (may exceed logcap: overflow)
    const uint32_t *qlist;    // optional: work item i is query qlist[i] (fallback pass), results other than cmps untouched
    uint2 *out_exp;           // build mode (graph construction phase 3): [nq][exp_cap] expanded (dist bits, id) in pop order
    uint32_t exp_cap, tgt_base;
    uint32_t *out_nexp;       // [nq] number of expansions
    uint32_t id_bits;         // VIS=1: ceil(log2(nd))
    const uint16_t *base_bf;  // fast mode (BF): bf16 copy of the base, rows padded to stride_bf elements (multiple of 128)
    uint32_t stride_bf;
#ifdef RG_K1_PROF
    unsigned long long *prof; // instrumented build only: [nq][16] per-phase cycle sums and event counts
#endif
};

#ifdef RG_K1_PROF
// instrumented build (make prof): s_memtime at the phase boundaries of a hop, summed per query
#define RG_PROF_DECL unsigned long long pf_t = clock64(), pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define RG_PROF(i) { const unsigned long long t_ = clock64(); pf_acc[i] += t_ - pf_t; pf_t = t_; }
#define RG_PROF_CNT(i, v) { pf_cnt[i] += (v); }
#else
#define RG_PROF_DECL
#define RG_PROF(i)
#define RG_PROF_CNT(i, v)
#endif

struct Beam {
    uint2 *ent;  // LDS: x = distance bits, y = id | kFlagBit
    uint32_t size, cur, cap;
};

// closest_unexpanded (neighbor.h:185-192): flag the entry at cur, move cur to the next unflagged entry
__device__ __forceinline__ uint2 beam_pop(Beam &bm, int lane) {
    uint2 e = bm.ent[bm.cur];
    if (lane == 0) bm.ent[bm.cur].y = e.y | kFlagBit;
    uint32_t c = bm.cur + 1;
    for (;;) {
        if (c >= bm.size) { c = bm.size; break; }
        uint32_t idx = c + lane;
        bool open = idx < bm.size && !(bm.ent[idx].y & kFlagBit);
        unsigned long long m = __ballot(open);
        if (m) { c += __ffsll((long long)m) - 1; break; }
        c += kWave;
    }
    bm.cur = c;
    wave_sync();
    return make_uint2(e.x, e.y & ~kFlagBit);
}

// Insert the n (<= 64) scored candidates (lane i holds candidate i) -- the net effect of n calls of
// NeighborPriorityQueue::insert (neighbor.h:150-183).  The beam is the top-cap of everything inserted so far
// under the total order (distance, id); candidates are distinct unvisited nodes, the only possible repeat is the
// entry point (never marked visited, index_bipartite.cpp:2349), whose second insert the reference drops.
template <bool DEDUP>
__device__ __forceinline__ void beam_merge(Beam &bm, float cd, uint32_t cid, uint32_t n, uint32_t ep, int lane) {
    bool valid = (uint32_t)lane < n && cid != ep;
    if (bm.size == bm.cap) {  // full: only candidates better than the current worst can enter (neighbor.h:151-153)
        uint2 w = bm.ent[bm.cap - 1];
        valid = valid && nb_less(cd, cid, __uint_as_float(w.x), w.y & ~kFlagBit);
    }
    if (!__any(valid)) return;
    // rank among the beam entries: lower bound under (distance, id)
    uint32_t lo = 0, hi = valid ? bm.size : 0;
    while (__any(lo < hi)) {
        if (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint2 e = bm.ent[mid];
            if (nb_less(__uint_as_float(e.x), e.y & ~kFlagBit, cd, cid)) lo = mid + 1;
            else hi = mid;
        }
    }
    if (DEDUP) {
        // With the lossy visited filter a node can be scored again.  Its distance bits are the same, so the lower bound
        // lands exactly on its beam entry if it is still there: drop it (the reference's equal-id rule, neighbor.h:161);
        // if it was evicted the tail test above already rejected it.  Same id twice in one hop: keep the lowest lane.
        if (valid && lo < bm.size && (bm.ent[lo].y & ~kFlagBit) == cid) valid = false;
        const unsigned long long m0 = __ballot(valid);
        bool dup = false;
        for (unsigned long long m = m0; m; m &= m - 1) {
            const int s = __ffsll((long long)m) - 1;
            dup = dup || (readlane_u(cid, s) == cid && s < lane);
        }
        valid = valid && !dup;
    }
    const unsigned long long vmask = __ballot(valid);
    if (!vmask) return;
    const uint32_t nc = __popcll(vmask);
    // rank among the candidates
    uint32_t crank = 0;
    for (unsigned long long m = vmask; m; m &= m - 1) {
        const int s = __ffsll((long long)m) - 1;
        const float od = readlane_f(cd, s);
        const uint32_t oi = readlane_u(cid, s);
        crank += nb_less(od, oi, cd, cid) ? 1u : 0u;
    }
    const uint32_t qrank = valid ? lo : 0xffffffffu;
    const uint32_t fpos = lo + crank;
    const bool keep = valid && fpos < bm.cap;
    // first beam index that moves
    const uint32_t minq = wave_min_u32(qrank);
    // new cursor: first unflagged entry after the merge
    uint32_t ncur = 0xffffffffu;
    if (bm.cur < bm.size) {
        const uint32_t sh = __popcll(__ballot(valid && qrank <= bm.cur));
        if (bm.cur + sh < bm.cap) ncur = bm.cur + sh;
    }
    ncur = min(ncur, wave_min_u32(keep ? fpos : 0xffffffffu));
    // shift entries [minq, size) right by the number of candidates ranked at or before them.  Done in place, top group
    // first; a group is up to 8 chunks of 64 entries held in registers, so its reads all complete before its writes
    // (which only land on indices >= the ones read, i.e. inside the group or in groups already moved).
    constexpr int G = 8;
    for (int top = (int)bm.size - 1; top >= (int)minq; top -= kWave * G) {
        uint2 e[G];
        uint32_t sh[G];
#pragma unroll
        for (int g2 = 0; g2 < G; ++g2) {
            const int i = top - kWave * g2 - lane;
            e[g2] = i >= (int)minq ? bm.ent[i] : make_uint2(0, 0);
            sh[g2] = 0;
        }
        for (unsigned long long m = vmask; m; m &= m - 1) {
            const int s = __ffsll((long long)m) - 1;
            const int q = (int)readlane_u(qrank, s);
#pragma unroll
            for (int g2 = 0; g2 < G; ++g2) sh[g2] += q <= top - kWave * g2 - lane ? 1u : 0u;
        }
        wave_sync();
#pragma unroll
        for (int g2 = 0; g2 < G; ++g2) {
            const int i = top - kWave * g2 - lane;
            if (i >= (int)minq && (uint32_t)i + sh[g2] < bm.cap) bm.ent[(uint32_t)i + sh[g2]] = e[g2];
        }
        wave_sync();
    }
    if (keep) bm.ent[fpos] = make_uint2(__float_as_uint(cd), cid);
    bm.size = min(bm.cap, bm.size + nc);
    bm.cur = ncur == 0xffffffffu ? bm.size : ncur;
    wave_sync();
}

// DIMC: 0 = any dimension (query staged in LDS), else the compile-time dimension (query in registers)
// BF:   opt-in fast mode, NOT parity: the traversal scores a bf16 copy of the base (4 instead of 7 HBM
//       lines per d = 200 evaluation); at the end the whole beam is re-scored with the exact fp32 routine and the k best
//       by exact (distance, id) are returned, so the reported distances are exact for the returned ids.
template <bool L2, bool ELL, int R, int VIS, int DIMC, bool BF>
__global__ void __launch_bounds__(64) rg_search_kernel(SearchParams P) {
    static_assert(!BF || (DIMC != 0 && ELL), "fast mode: compile-time dimension, ELL adjacency");
    constexpr int NB = (DIMC + 127) / 128;                                // fast mode: LDS-DMA instructions per bf16 row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int g = lane >> 4;
    // LDS carve (all offsets multiples of 16 B)
    float *stage = reinterpret_cast<float *>(smem);                       // R * stage_floats
    float *qv = stage + P.stage_total;                                    // dim (DIMC == 0 only)
    uint32_t *cand_id = reinterpret_cast<uint32_t *>(qv + (DIMC ? 0u : P.dim));   // 64
    float qr[DIMC ? (DIMC + 15) / 16 : 1];
    float qb[BF ? 8 * NB : 1];
    float *cand_d = reinterpret_cast<float *>(cand_id + kWave);           // 64
    Beam bm;
    bm.ent = reinterpret_cast<uint2 *>(cand_d + kWave);                   // L
    bm.cap = P.L;
    // VIS=1: lossy exact-match visited filter (direct mapped, 16-bit remainders of a bijective id hash)
    // id-log staging: ids are appended here and flushed to HBM 64 at a time (256-B aligned full-line stores; small
    // unaligned appends would turn into read-modify-writes at the memory side once the line has left L2)
    uint32_t *logbuf = reinterpret_cast<uint32_t *>(bm.ent + P.L);        // 128
    uint16_t *vtab = reinterpret_cast<uint16_t *>(logbuf + 128);
    const uint32_t vf_rem_bits = P.id_bits > P.vf_slots_log2 ? P.id_bits - P.vf_slots_log2 : 0u;
    const uint32_t vf_id_mask = P.id_bits >= 32u ? 0xffffffffu : ((1u << P.id_bits) - 1u);

    uint32_t *vmap = P.visited + (size_t)blockIdx.x * P.vwords;
    uint32_t epoch = VIS == 0 ? P.slot_epoch[blockIdx.x] : 0u;

    for (;;) {
        uint32_t qi = 0;
        if (lane == 0) qi = atomicAdd(P.counter, 1u);
        qi = readlane_u(qi, 0);
        if (qi >= P.nq) break;
        const bool cmps_only = P.qlist != nullptr;
        const bool build = P.out_exp != nullptr;
        const uint32_t tgt = P.tgt_base + qi;   // build mode: the node being linked is never scored (:1327)
        if (cmps_only) qi = P.qlist[qi];
        const float *query = P.queries + (size_t)qi * P.qstride;
        uint32_t *qlog = (VIS == 1 && P.qlog) ? P.qlog + (size_t)qi * P.logcap : nullptr;
        uint32_t logn = 0, lbn = 0;   // ids scored so far / ids waiting in logbuf
        RG_PROF_DECL;
        if constexpr (DIMC != 0) load_query_regs<DIMC>(query, qr, lane);
        else for (uint32_t i = lane; i < P.dim; i += kWave) qv[i] = query[i];
        if constexpr (BF) load_query_regs_bf<DIMC>(query, qb, lane);
        // new visited epoch (VisitedList::reset, visited_list_pool.h:20-26: ++curV, wipe on wrap)
        uint32_t etag = 0;
        if (VIS == 0) {
            if (++epoch == 0x10000u) {
                for (uint32_t w = lane; w < P.vwords; w += kWave) vmap[w] = 0u;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                epoch = 1;
            }
            etag = epoch << 16;
        }
        if (VIS == 1 || P.vf_front) {
            uint32_t *vt32 = reinterpret_cast<uint32_t *>(vtab);
            for (uint32_t i = lane; i < (1u << P.vf_slots_log2) / 2u; i += kWave) vt32[i] = 0xffffffffu;
        }
        wave_sync();

        // entry point: scored and queued, not marked visited (index_bipartite.cpp:2338-2352)
        // exact fp32 score of a staged pass / traversal score (the same thing unless BF)
        auto score_exact = [&](const float *buf) __attribute__((always_inline)) {
            if constexpr (DIMC != 0) return gather_score_q<L2, DIMC>(buf, qr, lane);
            else return gather_score<L2>(buf, qv, P.dim, lane);
        };
        auto score = [&](const float *buf) __attribute__((always_inline)) {
            if constexpr (BF) return score_bf<L2, NB>(reinterpret_cast<const uint32_t *>(buf), qb, lane);
            else return score_exact(buf);
        };
        auto issue = [&](uint32_t rid, bool act, float *buf) __attribute__((always_inline)) {
            if constexpr (BF) gather_issue_bf<NB>(P.base_bf + (size_t)rid * P.stride_bf, act, reinterpret_cast<uint32_t *>(buf), lane);
            else gather_issue(P.base + (size_t)rid * P.stride, P.dim, act, buf, lane);
        };
        issue(P.ep, g == 0, stage);
        gather_wait(0);
        const float epd = score(stage);
        if (lane == 0) bm.ent[0] = make_uint2(__float_as_uint(epd), P.ep);
        bm.size = 1;
        bm.cur = 0;
        wave_sync();

        uint32_t cmps = 0, hops = 0;
        RG_PROF(5);
        while (bm.cur < bm.size) {                                         // has_unexpanded_node, :2356
            const uint2 popped = beam_pop(bm, lane);                       // :2358
            RG_PROF(0);
#ifdef RG_K1_PROF
            const uint32_t pf_next = bm.cur < bm.size ? (bm.ent[bm.cur].y & ~kFlagBit) : 0xffffffffu;
            const uint32_t pf_cur0 = bm.cur;
#endif
            const uint32_t node = popped.y;
            if (build && lane == 0 && hops < P.exp_cap) P.out_exp[(size_t)qi * P.exp_cap + hops] = popped;   // full_retset, :1319
            ++hops;                                                        // :2366
            // adjacency of `node`, 64 words at a time
            uint32_t deg, first = 0;
            const uint32_t *list;
            if (ELL) {
                const uint32_t *row = P.ell + (size_t)node * P.ell_stride;
                first = (uint32_t)lane < P.ell_stride ? row[lane] : 0u;
                deg = readlane_u(first, 0);
                list = row + 1;
            } else {
                const uint64_t o0 = P.offsets[node], o1 = P.offsets[node + 1];
                deg = (uint32_t)(o1 - o0);
                list = P.nbrs + o0;
            }
#ifdef RG_K1_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            RG_PROF(1);
            for (uint32_t c0 = 0; c0 < deg; c0 += kWave) {                 // neighbour loop, :2368
                uint32_t id = 0;
                bool have;
                if (ELL && c0 == 0) {
                    // words 1..63 of the row were fetched with the degree: neighbours 0..62
                    id = (uint32_t)__shfl_down((int)first, 1, 64);
                    have = (uint32_t)lane < min(deg, 63u);
                    if (lane == 63 && deg > 63u) { id = list[63]; have = true; }
                } else {
                    have = c0 + lane < deg;
                    if (have) id = list[c0 + lane];
                }
                if (build && id == tgt) have = false;
                // visited test-and-set (:2378, :2385); same-hop duplicates are resolved by the atomic's order
                bool fresh = false;
                if (VIS == 1) {
                    // exact-match lookup: a hit proves "visited"; a miss is treated as fresh (may re-score a node whose
                    // entry was overwritten -- harmless for the beam, see beam_merge<true>)
                    if (have) {
                        const uint32_t x = (id * 0x9E3779B1u) & vf_id_mask;   // odd multiplier: bijection on id_bits bits
                        const uint32_t slot = x >> vf_rem_bits;
                        const uint16_t rem = (uint16_t)(x & ((1u << vf_rem_bits) - 1u));
                        fresh = vtab[slot] != rem;
                        if (fresh) vtab[slot] = rem;
                    }
                } else if (have && (P.diag & 1u)) fresh = true;
                else if (have && (P.diag & 2u)) {  // traffic without the dependency: fire-and-forget atomics
                    uint32_t *w = &vmap[id >> 4];
                    atomicMax(w, etag);
                    __hip_atomic_fetch_or(w, 1u << (id & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    fresh = true;
                } else if (have) {
                    // the LDS filter in front of the exact words: a hit proves "visited" and saves the two atomics (most
                    // repeat encounters on indexes with locality); a miss goes to the words, which decide
                    bool known = false;
                    if (P.vf_front) {
                        const uint32_t x = (id * 0x9E3779B1u) & vf_id_mask;
                        const uint32_t slot = x >> vf_rem_bits;
                        const uint16_t rem = (uint16_t)(x & ((1u << vf_rem_bits) - 1u));
                        known = vtab[slot] == rem;
                        if (!known) vtab[slot] = rem;
                    }
                    if (!known) {
                        uint32_t *w = &vmap[id >> 4];
                        const uint32_t bit = 1u << (id & 15u);
                        atomicMax(w, etag);                    // stale epoch -> word becomes (epoch, no bits)
                        const uint32_t old = atomicOr(w, bit); // same address, same lane: ordered behind the max
                        fresh = !(old & bit);
                    }
                }
                const unsigned long long fm = __ballot(fresh);
                const uint32_t n = __popcll(fm);
                RG_PROF_CNT(0, 1); RG_PROF_CNT(1, n);
                if (n == 0) { RG_PROF(2); continue; }
                if (fresh) {
                    const uint32_t slot = __popcll(fm & ((1ull << lane) - 1ull));
                    cand_id[slot] = id;
                    if (VIS == 1 && qlog) logbuf[lbn + slot] = id;
                }
                // a full 64-id line of the log leaves LDS here but is STORED after this hop's gathers have been consumed:
                // a store issued in front of them would sit at the head of the vmcnt queue and put its completion
                // latency on the critical path of the first counted wait
                bool flush = false;
                uint32_t flush_v = 0, flush_pos = 0;
                if (VIS == 1 && qlog) {
                    lbn += n;
                    if (lbn >= (uint32_t)kWave) {
                        lds_sync();
                        flush = true;
                        flush_pos = logn - (lbn - n);                     // ids already flushed (multiple of 64)
                        flush_v = logbuf[lane];
                        const uint32_t rest = logbuf[kWave + lane];
                        lds_sync();
                        lbn -= kWave;
                        if ((uint32_t)lane < lbn) logbuf[lane] = rest;
                    }
                }
                logn += n;
                cmps += n;                                                 // :2397
                wave_sync();
                RG_PROF(2);
                // gather + score (:2387): 4 rows per pass, a ring of R staging buffers keeps up to R passes in flight;
                // pass p is consumed once only the loads of the passes issued after it are still outstanding
                {
                    const uint32_t npass = (n + 3u) >> 2, lpp = BF ? (uint32_t)NB : loads_per_pass(P.dim);
                    for (uint32_t p = 0; p < (uint32_t)R && p < npass; ++p) {
                        const uint32_t c = 4 * p + g;
                        const uint32_t rid = c < n ? cand_id[c] : 0u;
                        issue(rid, c < n, stage + (size_t)p * P.stage_floats);
                    }
                    for (uint32_t p = 0; p < npass; ++p) {
                        const uint32_t last = min(npass, p + (uint32_t)R) - 1u;
                        if constexpr (DIMC != 0) {
                            constexpr int LPPC = BF ? NB : (DIMC + 63) / 64;
                            gather_wait_passes<LPPC>(last - p);
                        } else {
                            gather_wait((last - p) * lpp);
                        }
                        float *buf = stage + (size_t)(p & (R - 1)) * P.stage_floats;
                        const uint32_t c = 4 * p + g;
                        const float d = score(buf);
                        if (c < n && (lane & 15) == 0) cand_d[c] = d;
                        lds_sync();
                        if (p + R < npass) {
                            const uint32_t c2 = 4 * (p + R) + g;
                            const uint32_t rid = c2 < n ? cand_id[c2] : 0u;
                            issue(rid, c2 < n, buf);
                        }
                    }
                }
                if (VIS == 1 && flush && flush_pos + lane < P.logcap) qlog[flush_pos + lane] = flush_v;
                // queue inserts (:2398)
                const float cd = (uint32_t)lane < n ? cand_d[lane] : 0.0f;
                const uint32_t cid = (uint32_t)lane < n ? cand_id[lane] : 0u;
                wave_sync();
                RG_PROF(3);
#ifdef RG_K1_PROF
                { const uint32_t sz0 = bm.size; (void)sz0; }
#endif
                beam_merge<VIS == 1>(bm, cd, cid, n, P.ep, lane);
                RG_PROF(4);
            }
#ifdef RG_K1_PROF
            // would a speculative expansion of the next-to-pop node have been consumed?  (prediction = the entry that was
            // first unflagged right after the pop is still the first unflagged one after the merges)
            if (pf_next != 0xffffffffu) {
                RG_PROF_CNT(2, 1);
                const bool hit = bm.cur < bm.size && (bm.ent[bm.cur].y & ~kFlagBit) == pf_next;
                RG_PROF_CNT(3, hit ? 1 : 0);
                RG_PROF_CNT(4, bm.cur < pf_cur0 ? 1 : 0);   // cursor moved backwards: a candidate landed in front
            }
            RG_PROF_CNT(5, deg);
#endif
        }

        // results (:2408-2418)
        if (cmps_only || build) {
            if (build && lane == 0) P.out_nexp[qi] = hops;
        } else if (bm.size < P.k) {
            if (lane == 0) atomicMin(P.status, ((unsigned long long)(qi + P.qbase) << 32) | bm.size);
        } else if (BF) {
            // re-rank: exact fp32 distance of every beam entry (4 rows per pass through the exact routine), then the k
            // best by exact (distance, id), selected k times with a wave-wide minimum over an order-preserving key
            for (uint32_t i0 = 0; i0 < bm.size; i0 += 4) {
                const uint32_t i = i0 + g;
                const uint32_t rid = i < bm.size ? (bm.ent[i].y & ~kFlagBit) : 0u;
                gather_issue(P.base + (size_t)rid * P.stride, P.dim, i < bm.size, stage, lane);
                gather_wait(0);
                const float d = score_exact(stage);
                lds_sync();
                if (i < bm.size && (lane & 15) == 0) bm.ent[i] = make_uint2(__float_as_uint(d), rid);   // flag cleared
            }
            wave_sync();
            for (uint32_t r = 0; r < P.k; ++r) {
                uint32_t bh = 0xffffffffu, bl = 0xffffffffu, bi = 0xffffffffu;   // (ordered distance, id, beam index)
                for (uint32_t i = lane; i < bm.size; i += kWave) {
                    const uint2 e = bm.ent[i];
                    if (e.y & kFlagBit) continue;
                    const uint32_t u = e.x, o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                    if (o < bh || (o == bh && e.y < bl)) { bh = o; bl = e.y; bi = i; }
                }
                const uint32_t mh = wave_min_u32(bh);
                const uint32_t ml = wave_min_u32(bh == mh ? bl : 0xffffffffu);
                if (bh == mh && bl == ml && bi != 0xffffffffu) {   // ids are unique in the beam: exactly one lane
                    const uint2 e = bm.ent[bi];
                    bm.ent[bi].y = e.y | kFlagBit;
                    P.out_ids[(size_t)qi * P.k + r] = e.y;
                    P.out_dists[(size_t)qi * P.k + r] = __uint_as_float(e.x);
                }
                wave_sync();
            }
        } else {
            for (uint32_t i = lane; i < P.k; i += kWave) {
                const uint2 e = bm.ent[i];
                P.out_ids[(size_t)qi * P.k + i] = e.y & ~kFlagBit;
                P.out_dists[(size_t)qi * P.k + i] = __uint_as_float(e.x);
            }
        }
        if (VIS == 1 && qlog && lbn) {   // tail of the id log
            lds_sync();
            const uint32_t pos = logn - lbn;
            if ((uint32_t)lane < lbn && pos + lane < P.logcap) qlog[pos + lane] = logbuf[lane];
        }
#ifdef RG_K1_PROF
        RG_PROF(5);
        if (P.prof && lane == 0) {
            for (int i = 0; i < 8; ++i) { P.prof[(size_t)qi * 16 + i] = pf_acc[i]; P.prof[(size_t)qi * 16 + 8 + i] = pf_cnt[i]; }
        }
#endif
        if (lane == 0) {
            if (P.out_cmps) P.out_cmps[qi] = cmps;
            if (P.out_hops && !cmps_only) P.out_hops[qi] = hops;
            if (VIS == 1 && P.qlog_n) P.qlog_n[qi] = logn;
        }
        wave_sync();
    }
    if (VIS == 0 && lane == 0) P.slot_epoch[blockIdx.x] = epoch;
}

// K4: exact number of DISTINCT ids a query scored == the reference's cmps (every unvisited neighbour is scored exactly
// once there, index_bipartite.cpp:2378-2397).  The LDS visited filter of K1 may score a node twice; this pass counts the
// distinct ids of the query's log with an exact set held in LDS, one 1024-thread workgroup per query (queries handed out
// by an atomic counter).  The insert chain is latency bound, so the set is bucketed: one ds_read_b128 sees a whole
// bucket and only the chosen empty slot is CAS'd (linear probing had probe tails of dozens of slots, and a wave pays the
// longest of its 64 lanes).  Logs larger than one table are processed in hash partitions; queries whose log overflowed
// (or whose overflow area filled up) are listed for the exact fallback pass.
//
// HALF = true (id_bits - bucket_bits <= 15): slots hold 16-bit remainders of the bijective hash id * odd mod 2^id_bits,
// bucket = its top bits, 8 slots per 16-byte bucket, 0xffff = empty.  An id lives only in its home bucket; ids whose
// bucket is full go to an exact side table of full ids (T/8 words).  2T slots in 4T bytes: ~40k ids per pass at T = 2^15.
// HALF = false: 4 full ids per bucket, double hashing between buckets, 3T/4 ids per pass.
template <bool HALF>
__global__ void __launch_bounds__(1024) rg_distinct_kernel(const uint32_t *qlog, uint32_t logcap, const uint32_t *qlog_n,
                                                           uint32_t nq, uint32_t *out_cmps, uint32_t *ovf_list,
                                                           uint32_t *ovf_count, uint32_t *work, uint32_t tbits, uint32_t id_bits,
                                                           uint32_t qbase, unsigned long long *totals) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);      // T words = T/4 buckets
    __shared__ uint32_t s_cnt, s_fail, s_q;
    const uint32_t T = 1u << tbits, bbits = tbits - 2u, bmask = (1u << bbits) - 1u;
    const uint32_t OV = HALF ? T / 8u : 0u;                  // side table of full ids behind the buckets
    uint32_t *side = tab + T;
    const uint32_t cap = HALF ? (T / 4u) * 5u : (T / 4u) * 3u;
    const uint32_t rbits = id_bits - bbits, hmask = id_bits >= 32u ? 0xffffffffu : (1u << id_bits) - 1u;
    const int tid = threadIdx.x;
    for (;;) {
        if (tid == 0) { s_q = atomicAdd(work, 1u); s_cnt = 0; }
        __syncthreads();
        const uint32_t q = s_q;
        if (q >= nq) break;
        const uint32_t n = qlog_n[q];
        if (tid == 0) s_fail = n > logcap ? 1u : 0u;
        uint32_t mine = 0;
        if (n <= logcap && n > 0) {
            const uint32_t *log = qlog + (size_t)q * logcap;
            const uint32_t parts = (n + cap - 1) / cap;
            for (uint32_t p = 0; p < parts; ++p) {
                for (uint32_t i = tid * 4; i < T + OV; i += blockDim.x * 4)
                    *reinterpret_cast<uint4 *>(tab + i) = make_uint4(~0u, ~0u, ~0u, ~0u);
                __syncthreads();
                for (uint32_t i0 = tid; i0 < n; i0 += blockDim.x * 4u) {
                    uint32_t v[4];   // 4 independent loads in flight, then the inserts
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t i = i0 + (uint32_t)u * blockDim.x;
                        v[u] = i < n ? log[i] : 0xffffffffu;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t id = v[u];
                        if (id == 0xffffffffu) continue;
                        if (parts > 1 && ((id * 0x85EBCA6Bu) >> 16) % parts != p) continue;
                        if (HALF) {
                            const uint32_t h = (id * 0x9E3779B1u) & hmask;
                            const uint32_t b = h >> rbits, rem = h & ((1u << rbits) - 1u);
                            for (;;) {
                                const uint4 t = *reinterpret_cast<const uint4 *>(tab + 4u * b);
                                const uint32_t w4[4] = {t.x, t.y, t.z, t.w};
                                int e = 8;
                                bool found = false;
#pragma unroll
                                for (int k = 7; k >= 0; --k) {
                                    const uint32_t hv = (k & 1) ? w4[k >> 1] >> 16 : w4[k >> 1] & 0xffffu;
                                    found |= hv == rem;
                                    if (hv == 0xffffu) e = k;
                                }
                                if (found) break;
                                if (e == 8) {   // home bucket full: exact side table
                                    uint32_t slot = (id * 0x85EBCA6Bu) >> (32u - (tbits - 3u)), probes = 0;
                                    for (;;) {
                                        const uint32_t old = atomicCAS(&side[slot], 0xffffffffu, id);
                                        if (old == 0xffffffffu) { ++mine; break; }
                                        if (old == id) break;
                                        slot = (slot + 1u) & (OV - 1u);
                                        if (++probes >= OV) { s_fail = 1; break; }
                                    }
                                    break;
                                }
                                const uint32_t w = e < 2 ? t.x : e < 4 ? t.y : e < 6 ? t.z : t.w;
                                const uint32_t nw = (e & 1) ? (w & 0x0000ffffu) | (rem << 16) : (w & 0xffff0000u) | rem;
                                if (atomicCAS(&tab[4u * b + (uint32_t)(e >> 1)], w, nw) == w) { ++mine; break; }
                            }
                        } else {
                            const uint32_t h = id * 0x9E3779B1u;
                            uint32_t b = h >> (32u - bbits), probes = 0;
                            const uint32_t step = ((h >> 3) | 1u) & bmask;
                            for (;;) {
                                const uint4 t = *reinterpret_cast<const uint4 *>(tab + 4u * b);
                                if (t.x == id || t.y == id || t.z == id || t.w == id) break;
                                const int e = t.x == ~0u ? 0 : t.y == ~0u ? 1 : t.z == ~0u ? 2 : t.w == ~0u ? 3 : 4;
                                if (e == 4) {
                                    b = (b + step) & bmask;
                                    if (++probes > bmask) { s_fail = 1; break; }
                                    continue;
                                }
                                const uint32_t old = atomicCAS(&tab[4u * b + (uint32_t)e], 0xffffffffu, id);
                                if (old == 0xffffffffu) { ++mine; break; }
                                if (old == id) break;
                            }
                        }
                    }
                }
                __syncthreads();
            }
            for (int o = 32; o; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o, 64);
            if ((tid & 63) == 0) atomicAdd(&s_cnt, mine);
        }
        __syncthreads();
        if (tid == 0) {
            if (s_fail) ovf_list[atomicAdd(ovf_count, 1u)] = q + qbase;   // out_cmps is already offset; the list is global
            else {
                out_cmps[q] = s_cnt;
                atomicAdd(&totals[0], (unsigned long long)n);       // evaluations performed / distinct nodes: how much the
                atomicAdd(&totals[1], (unsigned long long)s_cnt);   // forgetful filter re-scored (search_wait looks at it)
            }
        }
        __syncthreads();
    }
}

// K1b: out[i] = compare(base[ids[i]], query) for n ids; one wave scores 4*R rows per pass
template <bool L2, int R>
__global__ void __launch_bounds__(64) rg_score_kernel(const float *__restrict__ base, uint32_t stride, uint32_t dim,
                                                      const float *__restrict__ query, const uint32_t *__restrict__ ids,
                                                      uint32_t n, float *__restrict__ out, uint32_t stage_floats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x, g = lane >> 4;
    float *stage = reinterpret_cast<float *>(smem);
    float *qv = stage + (size_t)R * stage_floats;
    for (uint32_t i = lane; i < dim; i += kWave) qv[i] = query[i];
    wave_sync();
    // each wave owns a contiguous run of passes (4 ids each) and streams them through a ring of R staging buffers
    const uint32_t npass_all = (n + 3u) >> 2;
    const uint32_t per = (npass_all + gridDim.x - 1) / gridDim.x;
    const uint32_t p_lo = min(npass_all, blockIdx.x * per), p_hi = min(npass_all, p_lo + per);
    const uint32_t npass = p_hi - p_lo, lpp = loads_per_pass(dim);
    auto issue = [&](uint32_t p, float *buf) {
        const uint32_t c = 4 * (p_lo + p) + g;
        const bool act = c < n;
        uint32_t rid = 0;
        if (act) rid = ids[c];
        gather_issue(base + (size_t)rid * stride, dim, act, buf, lane);
    };
    for (uint32_t p = 0; p < (uint32_t)R && p < npass; ++p) issue(p, stage + (size_t)p * stage_floats);
    for (uint32_t p = 0; p < npass; ++p) {
        const uint32_t last = min(npass, p + (uint32_t)R) - 1u;
        gather_wait((last - p) * lpp);
        float *buf = stage + (size_t)(p & (R - 1)) * stage_floats;
        const uint32_t c = 4 * (p_lo + p) + g;
        const float d = gather_score<L2>(buf, qv, dim, lane);
        if (c < n && (lane & 15) == 0) out[c] = d;
        lds_sync();
        if (p + R < npass) issue(p + R, buf);
    }
}

// fp32 base -> bf16 copy (round to nearest even), rows zero-padded to stride_bf elements
__global__ void rg_base_to_bf16_kernel(const float *__restrict__ base, uint32_t nd, uint32_t dim, uint32_t stride,
                                       uint16_t *__restrict__ out, uint32_t stride_bf) {
    const size_t total = (size_t)nd * stride_bf;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / stride_bf), c = (uint32_t)(i % stride_bf);
        uint16_t v = 0;
        if (c < dim) {
            const uint32_t u = __float_as_uint(base[(size_t)r * stride + c]);
            v = (u & 0x7f800000u) == 0x7f800000u ? (uint16_t)(u >> 16) : (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        }
        out[i] = v;
    }
}

// CSR -> ELL ([deg, ids...] per node at a fixed stride), one wave per node
__global__ void rg_csr_to_ell_kernel(const uint64_t *offsets, const uint32_t *nbrs, uint32_t nd, uint32_t *ell,
                                     uint32_t ell_stride) {
    const int lane = threadIdx.x & 63;
    const uint32_t wpb = blockDim.x / 64;
    for (uint32_t node = blockIdx.x * wpb + threadIdx.x / 64; node < nd; node += gridDim.x * wpb) {
        const uint64_t o0 = offsets[node];
        const uint32_t deg = (uint32_t)(offsets[node + 1] - o0);
        uint32_t *row = ell + (size_t)node * ell_stride;
        if (lane == 0) row[0] = deg;
        for (uint32_t j = lane; j < ell_stride - 1; j += 64) row[1 + j] = j < deg ? nbrs[o0 + j] : 0u;
    }
}

// max degree, max neighbour id, edge count check
__global__ void rg_graph_stats_kernel(const uint64_t *offsets, const uint32_t *nbrs, uint32_t nd, uint32_t *max_deg,
                                      uint32_t *max_id) {
    uint32_t md = 0, mi = 0;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < nd; i += nth) md = max(md, (uint32_t)(offsets[i + 1] - offsets[i]));
    const uint64_t ne = offsets[nd];
    for (size_t e = tid; e < ne; e += nth) mi = max(mi, nbrs[e]);
    for (int o = 32; o; o >>= 1) {
        md = max(md, (uint32_t)__shfl_xor((int)md, o, 64));
        mi = max(mi, (uint32_t)__shfl_xor((int)mi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(max_deg, md);
        atomicMax(max_id, mi);
    }
}

}  // namespace rg

// -------------------------------------------------------------------------------------------------- host
using rg::set_error;

#include "rg_index_struct.h"

namespace rg {

static rg_status pick_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_error(RG_ERR_DEVICE, "no HIP device visible: the gfx950 path cannot run (there is no CPU fallback)");
    if (device < 0 || device >= n) return set_error(RG_ERR_ARG, "device index out of range");
    RG_HIP(hipSetDevice(device));
    return RG_OK;
}

static rg_status finish_graph(rg_index *ix, const uint64_t *d_off, const uint32_t *d_nb) {
    // stats + validation
    DevBuf<uint32_t> stat_buf;
    RG_HIP(stat_buf.alloc(2));
    uint32_t *d_stat = stat_buf.p;
    RG_HIP(hipMemset(d_stat, 0, 8));
    hipLaunchKernelGGL(rg_graph_stats_kernel, dim3(2048), dim3(256), 0, 0, d_off, d_nb, ix->nd, d_stat, d_stat + 1);
    uint32_t st[2];
    RG_HIP(hipMemcpy(st, d_stat, 8, hipMemcpyDeviceToHost));
    uint64_t ne = 0;
    RG_HIP(hipMemcpy(&ne, d_off + ix->nd, 8, hipMemcpyDeviceToHost));
    ix->max_deg = st[0];
    ix->n_edges = ne;
    if (ne > 0 && st[1] >= ix->nd) return set_error(RG_ERR_FORMAT, "index file references a node id >= npts");
    if (ix->ep >= ix->nd) return set_error(RG_ERR_FORMAT, "entry point >= npts");
    if (ix->nd >= 0x80000000u) return set_error(RG_ERR_ARG, "more than 2^31-1 base points are not supported");
    // ELL (one load per hop instead of two dependent ones) when it costs at most 2.5x the CSR bytes (real RoarGraph
    // indexes: max degree <= 2*M_pjbp, avg ~ 0.6*max) or fits 16 GiB anyway (288 GB of HBM: a 10M-node index of maximum
    // degree 70 is 3.2 GB)
    const uint32_t es = (ix->max_deg + 1 + 15) / 16 * 16;
    const double ell_bytes = (double)ix->nd * es * 4.0, csr_bytes = (double)ne * 4.0 + (double)ix->nd * 8.0;
    if (!ix->force_csr && (ell_bytes <= 2.5 * csr_bytes + (64 << 20) || ell_bytes <= 16.0 * (1ull << 30))) {
        ix->ell_stride = es;
        RG_HIP(hipMalloc(&ix->d_ell, (size_t)ix->nd * es * 4));
        hipLaunchKernelGGL(rg_csr_to_ell_kernel, dim3(4096), dim3(256), 0, 0, d_off, d_nb, ix->nd, ix->d_ell, es);
        RG_HIP(hipDeviceSynchronize());
    } else {
        RG_HIP(hipMalloc(&ix->d_offsets, ((size_t)ix->nd + 1) * 8));
        RG_HIP(hipMalloc(&ix->d_nbrs, std::max<size_t>(ne * 4, 4)));
        RG_HIP(hipMemcpy(ix->d_offsets, d_off, ((size_t)ix->nd + 1) * 8, hipMemcpyDeviceToDevice));
        RG_HIP(hipMemcpy(ix->d_nbrs, d_nb, ne * 4, hipMemcpyDeviceToDevice));
    }
    RG_HIP(hipMalloc(&ix->d_counter, 64));
    RG_HIP(hipMalloc(&ix->d_status, 64));
    RG_HIP(hipHostMalloc(&ix->h_status, 64));
    RG_HIP(hipEventCreate(&ix->tune.ev0));
    RG_HIP(hipEventCreate(&ix->tune.ev1));
    hipDeviceProp_t prop;
    RG_HIP(hipGetDeviceProperties(&prop, ix->device));
    ix->num_cu = prop.multiProcessorCount;
    return RG_OK;
}

static uint32_t id_bits_of(uint32_t nd) {
    uint32_t b = 1;
    while (b < 32 && (1ull << b) < nd) ++b;
    return b;
}
static uint32_t filter_log2_of(const rg_index *ix) {  // remainder must fit 15 bits
    const uint32_t bits = id_bits_of(ix->nd);
    uint32_t t = (uint32_t)std::max(4, std::min(14, ix->filter_log2 > 0 ? ix->filter_log2 : ix->filter_auto));
    if (bits > t + 15) t = bits - 15;
    return std::min(t, bits);
}
// dimensions with a register-query instantiation of K1 (the BASELINE configs); ELL adjacency only
static int dimc_of(const rg_index *ix) {
    return (ix->d_ell != nullptr && (ix->dim == 200 || ix->dim == 512) && !ix->query_in_lds) ? (int)ix->dim : 0;
}
static bool fast_bf16_on(const rg_index *ix) { return ix->bf_launch; }   // decided per launch in launch_k1
static size_t stage_pass_floats(const rg_index *ix) {
    return fast_bf16_on(ix) ? (size_t)((ix->dim + 127) / 128) * 256 : (size_t)((ix->dim + 63) / 64) * 256;
}
static size_t stage_total_floats(const rg_index *ix, int R) {   // the fast mode's exact re-rank needs one fp32 pass
    return std::max((size_t)R * stage_pass_floats(ix), (size_t)((ix->dim + 63) / 64) * 256);
}
static size_t search_lds_bytes(const rg_index *ix, uint32_t L, int R) {
    size_t b = stage_total_floats(ix, R) * 4 + (dimc_of(ix) ? 0 : (size_t)ix->dim * 4) + 64 * 4 + 64 * 4 + (size_t)L * 8;
    if (ix->visited_mode != 0 || ix->exact_filter) b += 128 * 4 + std::max<size_t>(4, (size_t)2 << filter_log2_of(ix));
    return (b + 15) / 16 * 16;
}

static rg_status ensure_scratch(rg_index *ix, uint32_t slots) {
    const uint32_t vwords = (ix->nd + 15) / 16;
    if (ix->slots >= slots && ix->vwords == vwords) return RG_OK;
    if (ix->d_visited) (void)hipFree(ix->d_visited);
    if (ix->d_epoch) (void)hipFree(ix->d_epoch);
    ix->d_visited = nullptr;
    ix->d_epoch = nullptr;
    ix->slots = 0;
    RG_HIP(hipMalloc(&ix->d_visited, (size_t)slots * vwords * 4));
    RG_HIP(hipMemset(ix->d_visited, 0, (size_t)slots * vwords * 4));
    RG_HIP(hipMalloc(&ix->d_epoch, (size_t)slots * 4));
    RG_HIP(hipMemset(ix->d_epoch, 0, (size_t)slots * 4));
    ix->slots = slots;
    ix->vwords = vwords;
    return RG_OK;
}

template <bool L2, bool ELL, int R, int VIS, int DIMC, bool BF = false>
static rg_status launch_search_d(rg_index *ix, const SearchParams &P, uint32_t grid, size_t lds, hipStream_t s) {
    auto kern = rg_search_kernel<L2, ELL, R, VIS, DIMC, BF>;
    RG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, s, P);
    RG_HIP(hipGetLastError());
    return RG_OK;
}

template <bool L2, bool ELL, int R, int VIS>
static rg_status launch_search_v(rg_index *ix, const SearchParams &P, uint32_t grid, size_t lds, hipStream_t s) {
    if constexpr (ELL) {
        const int dc = dimc_of(ix);
        if constexpr (R <= 2) {
            if (P.base_bf && dc == 200) return launch_search_d<L2, ELL, R, VIS, 200, true>(ix, P, grid, lds, s);
            if (P.base_bf && dc == 512) return launch_search_d<L2, ELL, R, VIS, 512, true>(ix, P, grid, lds, s);
        }
        if (dc == 200) return launch_search_d<L2, ELL, R, VIS, 200>(ix, P, grid, lds, s);
        if (dc == 512) return launch_search_d<L2, ELL, R, VIS, 512>(ix, P, grid, lds, s);
    }
    return launch_search_d<L2, ELL, R, VIS, 0>(ix, P, grid, lds, s);
}

template <bool L2, bool ELL, int R>
static rg_status launch_search_t(rg_index *ix, const SearchParams &P, uint32_t grid, size_t lds, hipStream_t s) {
    return P.visited == nullptr ? launch_search_v<L2, ELL, R, 1>(ix, P, grid, lds, s)
                                 : launch_search_v<L2, ELL, R, 0>(ix, P, grid, lds, s);
}

template <bool L2, bool ELL>
static rg_status launch_search_r(rg_index *ix, const SearchParams &P, uint32_t grid, size_t lds, int R, hipStream_t s) {
    switch (R) {
        case 1: return launch_search_t<L2, ELL, 1>(ix, P, grid, lds, s);
        case 2: return launch_search_t<L2, ELL, 2>(ix, P, grid, lds, s);
        default: return launch_search_t<L2, ELL, 4>(ix, P, grid, lds, s);
    }
}

struct BuildOut { uint2_pod *exp; uint32_t exp_cap, node0; uint32_t *nexp; };
#ifdef RG_K1_PROF
static unsigned long long *g_prof_buf = nullptr;   // [nq][16], set through rg_prof_buffer (instrumented build only)
#endif

// one K1 launch.  mode: 0 exact HBM visited words, 1 LDS filter (optionally logging the scored ids).
static rg_status launch_k1(rg_index *ix, int mode, const float *d_q, uint32_t nq, uint32_t qstride, uint32_t k,
                           uint32_t L, uint32_t *d_ids, float *d_dists, uint32_t *d_cmps, uint32_t *d_hops,
                           const uint32_t *qlist, bool with_log, hipStream_t s, const BuildOut *bp = nullptr,
                           uint32_t qbase = 0) {
    // rows in flight per query: two passes of four pay on graphs with many fresh neighbours per hop (measured: +4 % at
    // out-degree 40, -4 % at 16, where the extra staging only costs resident queries)
    // A batch that leaves most wave slots empty is latency bound per query: LDS is plentiful then, so each query keeps
    // 16 rows in flight (two hop-latency round trips instead of five to ten).
    const int rpp = ix->rows_per_pass > 0 ? ix->rows_per_pass
                    : (nq <= (uint32_t)ix->num_cu * 6u && !bp) ? 16
                    : ((double)ix->n_edges >= 28.0 * ix->nd ? 8 : 4);
    int R = std::max(1, std::min(4, rpp / 4));
    if (R == 3) R = 2;
    // opt-in fast mode: plain top-k searches only (never the logging / recount / build launches)
    const bool bf = ix->fast_bf16 && ix->d_base_bf && dimc_of(ix) && !with_log && !bp && !qlist;
    ix->bf_launch = bf;
    if (bf) R = std::min(R, 2);
    const int saved_mode = ix->visited_mode;
    ix->visited_mode = mode;  // search_lds_bytes() looks at it
    // LDS visited filter, automatic size: the largest of 2^12 .. 2^9 entries that still leaves 14 resident queries per
    // CU (on genuine indexes a forgetful filter re-scores up to 50 % more nodes; past that point the lost residency
    // costs more than the repeats -- scripts/exp/filter_real.py)
    for (int f = 12; f >= 9; --f) {
        ix->filter_auto = f;
        if (f == 9 || ix->lds_per_cu / search_lds_bytes(ix, L, R) >= 14) break;
    }
    size_t lds = search_lds_bytes(ix, L, R);
    while (lds > ix->lds_per_cu && R > 1) { R >>= 1; lds = search_lds_bytes(ix, L, R); }
    ix->visited_mode = saved_mode;
    if (lds > ix->lds_per_cu) return set_error(RG_ERR_ARG, "L_pq too large for the 160 KiB LDS of one CU");
    int wpc = (int)std::min<size_t>(ix->lds_per_cu / lds, 32);
    if (ix->waves_per_cu > 0) wpc = std::min(wpc, ix->waves_per_cu);
    else wpc = std::min(wpc, 24);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(nq, (uint64_t)ix->num_cu * wpc);
    if (mode == 0) {
        rg_status st = ensure_scratch(ix, grid);
        if (st != RG_OK) return st;
    }
    RG_HIP(hipMemsetAsync(ix->d_counter, 0, 4, s));
    SearchParams P;
    P.base = ix->d_base; P.stride = ix->stride; P.dim = ix->dim; P.nd = ix->nd;
    P.ell = ix->d_ell; P.ell_stride = ix->ell_stride; P.offsets = ix->d_offsets; P.nbrs = ix->d_nbrs;
    P.ep = ix->ep; P.queries = d_q; P.nq = nq; P.qstride = qstride; P.k = k; P.L = L;
    P.out_ids = d_ids; P.out_dists = d_dists; P.out_cmps = d_cmps; P.out_hops = d_hops;
    P.visited = mode == 0 ? ix->d_visited : nullptr; P.vwords = ix->vwords; P.slot_epoch = ix->d_epoch;
    P.counter = ix->d_counter; P.status = ix->d_status;
    P.stage_floats = (uint32_t)stage_pass_floats(ix);
    P.stage_total = (uint32_t)stage_total_floats(ix, R);
    P.base_bf = bf ? ix->d_base_bf : nullptr; P.stride_bf = ix->stride_bf;
    P.diag = (uint32_t)ix->diag;
    P.qbase = qbase;
    P.vf_slots_log2 = filter_log2_of(ix);
    P.vf_front = (mode == 0 && ix->exact_filter) ? 1u : 0u;
    P.id_bits = id_bits_of(ix->nd);
    P.qlog = with_log ? ix->d_qlog : nullptr; P.logcap = ix->logcap; P.qlog_n = with_log ? ix->d_qlog_n : nullptr;
    P.qlist = qlist;
#ifdef RG_K1_PROF
    P.prof = (qlist || bp) ? nullptr : g_prof_buf;
#endif
    P.out_exp = nullptr; P.exp_cap = 0; P.tgt_base = 0; P.out_nexp = nullptr;
    if (bp) { P.out_exp = reinterpret_cast<uint2 *>(bp->exp); P.exp_cap = bp->exp_cap; P.tgt_base = bp->node0; P.out_nexp = bp->nexp; }
    const bool l2 = ix->metric == RG_METRIC_L2, ell = ix->d_ell != nullptr;
    if (l2 && ell) return launch_search_r<true, true>(ix, P, grid, lds, R, s);
    if (l2) return launch_search_r<true, false>(ix, P, grid, lds, R, s);
    if (ell) return launch_search_r<false, true>(ix, P, grid, lds, R, s);
    return launch_search_r<false, false>(ix, P, grid, lds, R, s);
}

static rg_status ensure_qlog(rg_index *ix, uint32_t nq) {
    // per-query id log: 128K ids (512 KiB) each; a batch larger than the log budget allows is searched in sub-batches
    // that reuse the same logs (search_dev).  Longer logs take the exact fallback pass.
    uint32_t cap = 1u << 17;
    if (ix->log_cap_knob > 0) cap = (uint32_t)ix->log_cap_knob;
    const size_t budget = (size_t)std::max(1, ix->log_budget_kb) << 10;
    const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(nq, budget / ((size_t)cap * 4)));
    if (ix->d_qlog && ix->qlog_nq >= chunk && ix->logcap == cap && ix->ovf_nq >= nq) { ix->qlog_chunk = chunk; return RG_OK; }
    if (ix->d_qlog) (void)hipFree(ix->d_qlog);
    if (ix->d_qlog_n) (void)hipFree(ix->d_qlog_n);
    if (ix->d_ovf) (void)hipFree(ix->d_ovf);
    ix->d_qlog = ix->d_qlog_n = ix->d_ovf = nullptr;
    ix->qlog_nq = ix->ovf_nq = 0;
    RG_HIP(hipMalloc(&ix->d_qlog, (size_t)chunk * cap * 4));
    RG_HIP(hipMalloc(&ix->d_qlog_n, (size_t)chunk * 4));
    RG_HIP(hipMalloc(&ix->d_ovf, ((size_t)nq + 2) * 4));   // [0] = count, [1] = K4 work counter, then the list
    ix->qlog_nq = chunk;
    ix->qlog_chunk = chunk;
    ix->ovf_nq = nq;
    ix->logcap = cap;
    return RG_OK;
}

static rg_status search_dev(rg_index *ix, const float *d_q, uint32_t nq, uint32_t qstride, uint32_t k, uint32_t L,
                            uint32_t *d_ids, float *d_dists, uint32_t *d_cmps, uint32_t *d_hops, hipStream_t s) {
    if (!ix) return set_error(RG_ERR_ARG, "null index");
    if (k > L) return set_error(RG_ERR_ARG, "L_pq must greater or equal than k");  // test_search_roargraph.cpp:192-195
    if (k == 0 || L == 0) return set_error(RG_ERR_ARG, "k and L_pq must be positive");
    if (qstride < ix->dim) return set_error(RG_ERR_ARG, "query stride smaller than the index dimension");
    if (nq == 0) return RG_OK;
    RG_HIP(hipSetDevice(ix->device));
    RG_HIP(hipMemsetAsync(ix->d_status, 0xff, 8, s));
    ix->pending.active = false;
    const bool fast = ix->fast_bf16 && ix->d_base_bf && dimc_of(ix);
    bool exact_count = ix->visited_mode == 2 && d_cmps != nullptr && !fast;
    // fast mode under the default visited mode: the exact words from the beam width on at which the parity batches (if
    // there were any) found them faster, the LDS filter alone below it; "visited" 0 / 1 force one or the other
    if (fast && ix->visited_mode == 2 && L >= ix->exact_from_L)
        return launch_k1(ix, 0, d_q, nq, qstride, k, L, d_ids, d_dists, d_cmps, d_hops, nullptr, false, s);
    // Adaptive default: both exact forms return the same bits.  When a batch showed the LDS filter re-scoring nodes
    // wholesale at this beam width (performed > 1.3 x distinct: long searches on indexes with locality), the next batch
    // of that width runs on the exact HBM words as a timed trial; the faster form is kept from that width on (which one
    // wins depends on the index: the words of a 10M-node index are 10 GB of random atomics, those of a 2M-node index
    // mostly cache resident -- scripts/exp/visited_modes_real.py).
    ix->tune.timed = false;
    if (exact_count && ix->filter_log2 <= 0 && ix->tune.ev0) {
        const bool trial = ix->tune.trial_L == L && nq >= 1000;
        ix->tune.timed = true; ix->tune.mode = (L >= ix->exact_from_L || trial) ? 0 : 2; ix->tune.L = L; ix->tune.nq = nq;
        ix->tune.is_trial = trial && L < ix->exact_from_L;
        RG_HIP(hipEventRecord(ix->tune.ev0, s));
        if (ix->tune.mode == 0) {
            rg_status st0 = launch_k1(ix, 0, d_q, nq, qstride, k, L, d_ids, d_dists, d_cmps, d_hops, nullptr, false, s);
            if (st0 == RG_OK) RG_HIP(hipEventRecord(ix->tune.ev1, s));
            return st0;
        }
    }
    if (exact_count) RG_HIP(hipMemsetAsync(ix->d_status + 1, 0, 16, s));
    if (!exact_count)
        return launch_k1(ix, ix->visited_mode == 0 ? 0 : 1, d_q, nq, qstride, k, L, d_ids, d_dists, d_cmps, d_hops, nullptr, false, s);
    // mode 2: LDS-filter search with id log, then the exact distinct count (K4); overflowed logs are re-counted by an
    // exact pass inside rg_search_wait
    rg_status st = ensure_qlog(ix, nq);
    if (st != RG_OK) return st;
    RG_HIP(hipMemsetAsync(ix->d_ovf, 0, 8, s));
    // default table: 2^15 words = 128 KiB of LDS (+ 16 KiB side table in the half-word form)
    const uint32_t tbits = (uint32_t)std::max(6, std::min(15, ix->count_table_log2));
    const uint32_t bbits = tbits - 2u;
    const uint32_t id_bits = std::max(id_bits_of(ix->nd), bbits + 1u);
    const bool half = id_bits - bbits <= 15u && !ix->count_full_ids;
    const size_t lds = half ? ((size_t)4 << tbits) + ((size_t)4 << (tbits - 3u)) : (size_t)4 << tbits;
    for (uint32_t q0 = 0; q0 < nq; q0 += ix->qlog_chunk) {
        const uint32_t nqc = std::min(ix->qlog_chunk, nq - q0);
        if (q0) RG_HIP(hipMemsetAsync(ix->d_ovf + 1, 0, 4, s));   // K4 work counter; the overflow count keeps running
        st = launch_k1(ix, 1, d_q + (size_t)q0 * qstride, nqc, qstride, k, L, d_ids ? d_ids + (size_t)q0 * k : nullptr,
                       d_dists ? d_dists + (size_t)q0 * k : nullptr, d_cmps + q0, d_hops ? d_hops + q0 : nullptr, nullptr, true, s,
                       nullptr, q0);
        if (st != RG_OK) return st;
        const dim3 grid(std::min<uint32_t>(nqc, (uint32_t)ix->num_cu));
        if (half) {
            auto kern = rg_distinct_kernel<true>;
            RG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, grid, dim3(1024), lds, s, ix->d_qlog, ix->logcap, ix->d_qlog_n, nqc, d_cmps + q0, ix->d_ovf + 2,
                               ix->d_ovf, ix->d_ovf + 1, tbits, id_bits, q0, ix->d_status + 1);
        } else {
            auto kern = rg_distinct_kernel<false>;
            RG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, grid, dim3(1024), lds, s, ix->d_qlog, ix->logcap, ix->d_qlog_n, nqc, d_cmps + q0, ix->d_ovf + 2,
                               ix->d_ovf, ix->d_ovf + 1, tbits, id_bits, q0, ix->d_status + 1);
        }
    }
    RG_HIP(hipGetLastError());
    if (ix->tune.timed) RG_HIP(hipEventRecord(ix->tune.ev1, s));
    ix->pending.active = true;
    ix->pending.q = d_q; ix->pending.nq = nq; ix->pending.qstride = qstride; ix->pending.k = k; ix->pending.L = L;
    ix->pending.ids = d_ids; ix->pending.dists = d_dists; ix->pending.cmps = d_cmps; ix->pending.hops = d_hops;
    return RG_OK;
}

static rg_status search_wait(rg_index *ix, hipStream_t s, uint32_t k) {
    RG_HIP(hipMemcpyAsync(ix->h_status, ix->d_status, 8, hipMemcpyDeviceToHost, s));
    if (ix->pending.active) {
        RG_HIP(hipMemcpyAsync(ix->h_status + 1, ix->d_ovf, 4, hipMemcpyDeviceToHost, s));
        RG_HIP(hipMemcpyAsync(ix->h_status + 2, ix->d_status + 1, 16, hipMemcpyDeviceToHost, s));
    }
    RG_HIP(hipStreamSynchronize(s));
    const unsigned long long v = *ix->h_status;
    float per_q = 0.0f;   // time per query of the batch just finished (adaptive default only)
    if (ix->tune.timed) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ix->tune.ev0, ix->tune.ev1) == hipSuccess && ix->tune.nq) per_q = ms / (float)ix->tune.nq;
        ix->tune.timed = false;
        if (ix->tune.mode == 0 && ix->tune.is_trial && per_q > 0.0f) {   // verdict of the trial
            if (per_q < 0.97f * ix->tune.filter_per_q) ix->exact_from_L = std::min(ix->exact_from_L, ix->tune.L);
            else ix->tune.filter_ok_upto = std::max(ix->tune.filter_ok_upto, ix->tune.L);
            ix->tune.trial_L = 0;
        }
    }
    if (ix->pending.active) {
        ix->pending.active = false;
        const unsigned long long performed = ix->h_status[2], distinct = ix->h_status[3];
        if (distinct > 0 && (double)performed > 1.3 * (double)distinct && per_q > 0.0f && ix->pending.nq >= 1000 &&
            ix->pending.L > ix->tune.filter_ok_upto && ix->pending.L < ix->exact_from_L) {
            ix->tune.trial_L = ix->pending.L;      // next batch of this width: the exact words, timed
            ix->tune.filter_per_q = per_q;
        }
        const uint32_t novf = (uint32_t)(ix->h_status[1] & 0xffffffffu);
        if (novf > 0 && v == ~0ull) {
            // logs that did not fit: recount those queries with the exact HBM visited words (only cmps is rewritten)
            const auto &pd = ix->pending;
            rg_status st = launch_k1(ix, 0, pd.q, novf, pd.qstride, pd.k, pd.L, pd.ids, pd.dists, pd.cmps, pd.hops, ix->d_ovf + 2, false, s);
            if (st != RG_OK) return st;
            RG_HIP(hipStreamSynchronize(s));
        }
    }
    if (v != ~0ull) {
        char buf[160];
        if (k) snprintf(buf, sizeof buf, "not enough results: %u, expected: %u (query %u)", (unsigned)(v & 0xffffffffu), k, (unsigned)(v >> 32));
        else snprintf(buf, sizeof buf, "not enough results: %u (query %u)", (unsigned)(v & 0xffffffffu), (unsigned)(v >> 32));
        return set_error(RG_ERR_NOT_ENOUGH, buf);
    }
    return RG_OK;
}

static rg_status score_dev(rg_index *ix, const float *d_query, const uint32_t *d_ids, uint32_t n, float *d_out, hipStream_t s) {
    if (!ix) return set_error(RG_ERR_ARG, "null index");
    if (n == 0) return RG_OK;
    RG_HIP(hipSetDevice(ix->device));
    constexpr int R = 2;
    const uint32_t stage_floats = ((ix->dim + 63) / 64) * 256;
    const size_t lds = (size_t)R * stage_floats * 4 + (size_t)ix->dim * 4;
    const uint32_t passes = (n + 3) / 4;
    const uint32_t grid = std::min<uint32_t>(passes, (uint32_t)ix->num_cu * 16u);
    if (ix->metric == RG_METRIC_L2)
        hipLaunchKernelGGL((rg_score_kernel<true, R>), dim3(grid), dim3(64), lds, s, ix->d_base, ix->stride, ix->dim, d_query, d_ids, n, d_out, stage_floats);
    else
        hipLaunchKernelGGL((rg_score_kernel<false, R>), dim3(grid), dim3(64), lds, s, ix->d_base, ix->stride, ix->dim, d_query, d_ids, n, d_out, stage_floats);
    RG_HIP(hipGetLastError());
    return RG_OK;
}

rg_status build_search_dev(rg_index *ix, uint32_t node0, uint32_t n, uint32_t L, uint2_pod *d_exp, uint32_t exp_cap,
                           uint32_t *d_nexp, void *stream) {
    if (!ix || !d_exp || !d_nexp) return set_error(RG_ERR_ARG, "null argument");
    if (n == 0) return RG_OK;
    if ((uint64_t)node0 + n > ix->nd) return set_error(RG_ERR_ARG, "node range out of bounds");
    RG_HIP(hipSetDevice(ix->device));
    RG_HIP(hipMemsetAsync(ix->d_status, 0xff, 8, (hipStream_t)stream));
    BuildOut bo{d_exp, exp_cap, node0, d_nexp};
    // queries are the base rows themselves; k = 1 (no top-k is written in build mode)
    return launch_k1(ix, 1, ix->d_base + (size_t)node0 * ix->stride, n, ix->stride, 1, L, nullptr, nullptr, nullptr, nullptr,
                     nullptr, false, (hipStream_t)stream, &bo);
}

rg_status build_index_create(const float *d_base, uint32_t nd, uint32_t dim, uint32_t stride, uint32_t ep, int metric,
                             int device, uint32_t ell_stride, rg_index **out) {
    rg_status st = pick_device(device);
    if (st != RG_OK) return st;
    if (dim == 0 || dim % 8 || stride < dim || stride % 4 || ((uintptr_t)d_base & 15))
        return set_error(RG_ERR_ARG, "device base must be 16-byte aligned with dim % 8 == 0 and stride % 4 == 0");
    rg_index *ix = new rg_index();
    ix->device = device; ix->metric = metric; ix->nd = nd; ix->dim = dim; ix->stride = stride; ix->ep = ep;
    ix->d_base = const_cast<float *>(d_base);
    ix->ell_stride = ell_stride;
    ix->max_deg = ell_stride - 1;
    hipError_t e = hipMalloc(&ix->d_ell, (size_t)nd * ell_stride * 4);
    if (e == hipSuccess) e = hipMemset(ix->d_ell, 0, (size_t)nd * ell_stride * 4);
    if (e == hipSuccess) e = hipMalloc(&ix->d_counter, 64);
    if (e == hipSuccess) e = hipMalloc(&ix->d_status, 64);
    if (e == hipSuccess) e = hipHostMalloc(&ix->h_status, 64);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { rg_index_close(ix); return set_error(RG_ERR_DEVICE, hipGetErrorString(e)); }
    ix->num_cu = prop.multiProcessorCount;
    *out = ix;
    return RG_OK;
}

rg_status build_index_set_ell(rg_index *ix, const uint32_t *h_ell, void *stream) {
    RG_HIP(hipSetDevice(ix->device));
    RG_HIP(hipMemcpyAsync(ix->d_ell, h_ell, (size_t)ix->nd * ix->ell_stride * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
    return RG_OK;
}

}  // namespace rg

extern "C" {

#ifdef RG_K1_PROF
void rg_prof_buffer(void *d_buf) { rg::g_prof_buf = static_cast<unsigned long long *>(d_buf); }
#endif

int rg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void rg_index_close(rg_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->own_base && ix->d_base) (void)hipFree(ix->d_base);
    if (ix->d_offsets) (void)hipFree(ix->d_offsets);
    if (ix->d_nbrs) (void)hipFree(ix->d_nbrs);
    if (ix->d_ell) (void)hipFree(ix->d_ell);
    if (ix->d_visited) (void)hipFree(ix->d_visited);
    if (ix->d_epoch) (void)hipFree(ix->d_epoch);
    if (ix->d_qlog) (void)hipFree(ix->d_qlog);
    if (ix->d_qlog_n) (void)hipFree(ix->d_qlog_n);
    if (ix->d_base_bf) (void)hipFree(ix->d_base_bf);
    if (ix->d_ovf) (void)hipFree(ix->d_ovf);
    if (ix->d_counter) (void)hipFree(ix->d_counter);
    if (ix->d_status) (void)hipFree(ix->d_status);
    if (ix->h_status) (void)hipHostFree(ix->h_status);
    if (ix->tune.ev0) (void)hipEventDestroy(ix->tune.ev0);
    if (ix->tune.ev1) (void)hipEventDestroy(ix->tune.ev1);
    delete ix;
}

rg_status rg_index_open_dev(const float *d_base, uint32_t nd, uint32_t dim, uint32_t stride, const uint64_t *d_offsets,
                            const uint32_t *d_nbrs, uint32_t ep, int metric, int device, rg_index **out) {
    if (!out || !d_base || !d_offsets) return set_error(RG_ERR_ARG, "null argument");
    if (metric != RG_METRIC_L2 && metric != RG_METRIC_IP && metric != RG_METRIC_COSINE)
        return set_error(RG_ERR_ARG, "Unknown distance type");
    if (dim == 0 || dim % 8 != 0 || stride < dim || stride % 4 != 0 || ((uintptr_t)d_base & 15))
        return set_error(RG_ERR_ARG, "device base must be 16-byte aligned with dim % 8 == 0 and stride % 4 == 0");
    rg_status st = rg::pick_device(device);
    if (st != RG_OK) return st;
    rg_index *ix = new rg_index();
    ix->device = device; ix->metric = metric; ix->nd = nd; ix->dim = dim; ix->stride = stride; ix->ep = ep;
    ix->d_base = const_cast<float *>(d_base);
    ix->own_base = false;
    if (const char *e = getenv("RG_FORCE_CSR")) ix->force_csr = atoi(e);
    st = rg::finish_graph(ix, d_offsets, d_nbrs);
    if (st != RG_OK) { rg_index_close(ix); return st; }
    *out = ix;
    return RG_OK;
}

rg_status rg_index_open_mem(const float *base, uint32_t nd, uint32_t dim, uint32_t stride, const uint64_t *offsets,
                            const uint32_t *nbrs, uint32_t ep, int metric, int device, rg_index **out) {
    if (!out || !base || !offsets) return set_error(RG_ERR_ARG, "null argument");
    if (dim == 0 || stride < dim) return set_error(RG_ERR_ARG, "bad dim/stride");
    rg_status st = rg::pick_device(device);
    if (st != RG_OK) return st;
    // device copy at the aligned stride, zero padded (data_align, util.h:37-75); cosine rows are normalised first
    const uint32_t ad = rg::aligned_dim(dim);
    rg::DevBuf<float> d_base;
    RG_HIP(d_base.alloc((size_t)nd * ad));
    if (metric == RG_METRIC_COSINE || ad != dim || ad != stride) {
        std::vector<float> tmp((size_t)nd * ad, 0.0f);
        for (size_t i = 0; i < nd; ++i) std::memcpy(tmp.data() + i * ad, base + i * (size_t)stride, (size_t)dim * 4);
        if (metric == RG_METRIC_COSINE) rg_normalize_rows(tmp.data(), nd, ad, dim);
        RG_HIP(hipMemcpy(d_base.p, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
    } else {
        RG_HIP(hipMemcpy(d_base.p, base, (size_t)nd * ad * 4, hipMemcpyHostToDevice));
    }
    const uint64_t ne = offsets[nd];
    rg::DevBuf<uint64_t> d_off;
    rg::DevBuf<uint32_t> d_nb;
    RG_HIP(d_off.alloc((size_t)nd + 1));
    RG_HIP(d_nb.alloc(ne));
    RG_HIP(hipMemcpy(d_off.p, offsets, ((size_t)nd + 1) * 8, hipMemcpyHostToDevice));
    RG_HIP(hipMemcpy(d_nb.p, nbrs, ne * 4, hipMemcpyHostToDevice));
    rg_index *ix = nullptr;
    st = rg_index_open_dev(d_base.p, nd, ad, ad, d_off.p, d_nb.p, ep, metric, device, &ix);
    if (st != RG_OK) return st;
    ix->own_base = true;
    (void)d_base.release();   // now owned by the index
    *out = ix;
    return RG_OK;
}

rg_status rg_index_open(const char *base_fbin, const char *index_path, int metric, int device, rg_index **out) {
    if (!base_fbin || !index_path || !out) return set_error(RG_ERR_ARG, "null argument");
    uint32_t nb = 0, dim = 0, stride = 0, nd = 0, ep = 0;
    float *base = nullptr;
    uint64_t *off = nullptr;
    uint32_t *nbrs = nullptr;
    rg_status st = rg_fbin_load(base_fbin, &nb, &dim, &stride, &base);
    if (st != RG_OK) return st;
    st = rg_graph_load(index_path, &nd, &ep, &off, &nbrs);
    if (st != RG_OK) { rg_free(base); return st; }
    if (nd != nb) {
        rg_free(base); rg_free(off); rg_free(nbrs);
        return set_error(RG_ERR_FORMAT, "index and base file disagree on the number of points");
    }
    st = rg_index_open_mem(base, nb, dim, stride, off, nbrs, ep, metric, device, out);
    rg_free(base); rg_free(off); rg_free(nbrs);
    return st;
}

rg_status rg_index_info(const rg_index *ix, uint32_t *nd, uint32_t *dim, uint32_t *stride, uint32_t *ep,
                        float *avg_degree, uint32_t *max_degree, int *device) {
    if (!ix) return set_error(RG_ERR_ARG, "null index");
    if (nd) *nd = ix->nd;
    if (dim) *dim = ix->dim;
    if (stride) *stride = ix->stride;
    if (ep) *ep = ix->ep;
    if (avg_degree) *avg_degree = ix->nd ? (float)((double)ix->n_edges / ix->nd) : 0.0f;
    if (max_degree) *max_degree = ix->max_deg;
    if (device) *device = ix->device;
    return RG_OK;
}

rg_status rg_index_set(rg_index *ix, const char *name, int value) {
    if (!ix || !name) return set_error(RG_ERR_ARG, "null argument");
    if (!strcmp(name, "waves_per_cu")) ix->waves_per_cu = value;
    else if (!strcmp(name, "rows_per_pass")) ix->rows_per_pass = value;
    else if (!strcmp(name, "diag")) ix->diag = value;
    else if (!strcmp(name, "visited")) ix->visited_mode = value < 0 || value > 2 ? 2 : value;
    else if (!strcmp(name, "filter_log2")) ix->filter_log2 = value;
    else if (!strcmp(name, "log_cap")) ix->log_cap_knob = value;
    else if (!strcmp(name, "log_budget_kb")) ix->log_budget_kb = value;
    else if (!strcmp(name, "query_in_lds")) ix->query_in_lds = value != 0;
    else if (!strcmp(name, "exact_filter")) ix->exact_filter = value != 0;
    else if (!strcmp(name, "fast_bf16")) {
        // opt-in, NOT parity (see rg.h): the bf16 copy of the base is made on first use
        if (value && !ix->d_base_bf) {
            if (hipSetDevice(ix->device) != hipSuccess) return set_error(RG_ERR_DEVICE, "cannot select the index device");
            ix->stride_bf = (ix->dim + 127u) / 128u * 128u;
            RG_HIP(hipMalloc(&ix->d_base_bf, (size_t)ix->nd * ix->stride_bf * 2));
            hipLaunchKernelGGL(rg::rg_base_to_bf16_kernel, dim3(ix->num_cu * 8), dim3(256), 0, 0, ix->d_base, ix->nd, ix->dim, ix->stride,
                               ix->d_base_bf, ix->stride_bf);
            RG_HIP(hipGetLastError());
            RG_HIP(hipDeviceSynchronize());
        }
        ix->fast_bf16 = value != 0;
    }
    else if (!strcmp(name, "count_table_log2")) ix->count_table_log2 = value;
    else if (!strcmp(name, "count_full_ids")) ix->count_full_ids = value != 0;
    else return set_error(RG_ERR_ARG, "unknown knob");
    return RG_OK;
}

rg_status rg_search_dev(rg_index *ix, const float *d_queries, uint32_t nq, uint32_t qstride, uint32_t k, uint32_t L_pq,
                        uint32_t *d_ids, float *d_dists, uint32_t *d_cmps, uint32_t *d_hops, void *stream) {
    return rg::search_dev(ix, d_queries, nq, qstride, k, L_pq, d_ids, d_dists, d_cmps, d_hops, (hipStream_t)stream);
}

rg_status rg_search_wait(rg_index *ix, void *stream) {
    if (!ix) return set_error(RG_ERR_ARG, "null index");
    return rg::search_wait(ix, (hipStream_t)stream, 0);
}

rg_status rg_search(rg_index *ix, const float *queries, uint32_t nq, uint32_t qstride, uint32_t k, uint32_t L_pq,
                    uint32_t *out_ids, float *out_dists, uint32_t *out_cmps, uint32_t *out_hops) {
    if (!ix || !queries || !out_ids || !out_dists) return set_error(RG_ERR_ARG, "null argument");
    if (k > L_pq) return set_error(RG_ERR_ARG, "L_pq must greater or equal than k");
    if (nq == 0) return RG_OK;
    RG_HIP(hipSetDevice(ix->device));
    // queries -> device at the index stride, zero padded; cosine queries normalised (test_search_roargraph.cpp:167-172)
    const uint32_t d = ix->dim;
    const uint32_t use = std::min(qstride, d);
    std::vector<float> hq((size_t)nq * d, 0.0f);
    for (size_t i = 0; i < nq; ++i) std::memcpy(hq.data() + i * d, queries + i * (size_t)qstride, (size_t)use * 4);
    if (ix->metric == RG_METRIC_COSINE) rg_normalize_rows(hq.data(), nq, d, d);
    rg::DevBuf<float> d_q, d_dist;
    rg::DevBuf<uint32_t> d_ids, d_ch;
    RG_HIP(d_q.alloc(hq.size()));
    RG_HIP(d_ids.alloc((size_t)nq * k));
    RG_HIP(d_dist.alloc((size_t)nq * k));
    RG_HIP(d_ch.alloc((size_t)nq * 2));
    RG_HIP(hipMemcpy(d_q.p, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    RG_HIP(hipMemset(d_ids.p, 0, (size_t)nq * k * 4));
    RG_HIP(hipMemset(d_dist.p, 0, (size_t)nq * k * 4));
    rg_status st = rg::search_dev(ix, d_q.p, nq, d, k, L_pq, d_ids.p, d_dist.p, d_ch.p, d_ch.p + nq, nullptr);
    if (st == RG_OK) st = rg::search_wait(ix, nullptr, k);
    if (st == RG_OK || st == RG_ERR_NOT_ENOUGH) {
        (void)hipMemcpy(out_ids, d_ids.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(out_dists, d_dist.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost);
        if (out_cmps) (void)hipMemcpy(out_cmps, d_ch.p, (size_t)nq * 4, hipMemcpyDeviceToHost);
        if (out_hops) (void)hipMemcpy(out_hops, d_ch.p + nq, (size_t)nq * 4, hipMemcpyDeviceToHost);
    }
    return st;
}

rg_status rg_search_sharded(rg_index *const *replicas, int nrep, const float *queries, uint32_t nq, uint32_t qstride,
                            uint32_t k, uint32_t L_pq, uint32_t *out_ids, float *out_dists, uint32_t *out_cmps,
                            uint32_t *out_hops) {
    if (!replicas || nrep <= 0 || !queries || !out_ids || !out_dists) return set_error(RG_ERR_ARG, "null argument");
    for (int r = 0; r < nrep; ++r) {
        if (!replicas[r]) return set_error(RG_ERR_ARG, "null replica");
        if (replicas[r]->dim != replicas[0]->dim || replicas[r]->nd != replicas[0]->nd || replicas[r]->metric != replicas[0]->metric)
            return set_error(RG_ERR_ARG, "replicas describe different indexes");
    }
    if (k > L_pq) return set_error(RG_ERR_ARG, "L_pq must greater or equal than k");
    if (nq == 0) return RG_OK;
    const uint32_t d = replicas[0]->dim, use = std::min(qstride, d);
    std::vector<float> hq((size_t)nq * d, 0.0f);
    for (size_t i = 0; i < nq; ++i) std::memcpy(hq.data() + i * d, queries + i * (size_t)qstride, (size_t)use * 4);
    if (replicas[0]->metric == RG_METRIC_COSINE) rg_normalize_rows(hq.data(), nq, d, d);
    struct Shard {   // buffers and stream of one replica's slice, released on every exit path
        int dev = 0;
        uint32_t lo = 0, n = 0;
        float *q = nullptr, *dist = nullptr;
        uint32_t *ids = nullptr, *ch = nullptr;
        hipStream_t s = nullptr;
        ~Shard() {
            if (!q && !dist && !ids && !ch && !s) return;
            (void)hipSetDevice(dev);
            if (s) (void)hipStreamSynchronize(s);
            (void)hipFree(q); (void)hipFree(dist); (void)hipFree(ids); (void)hipFree(ch);
            if (s) (void)hipStreamDestroy(s);
        }
    };
    std::vector<Shard> S((size_t)nrep);
    const uint32_t per = (nq + (uint32_t)nrep - 1) / (uint32_t)nrep;
    rg_status st = RG_OK;
    for (int r = 0; r < nrep && st == RG_OK; ++r) {
        Shard &sh = S[(size_t)r];
        rg_index *ix = replicas[r];
        sh.dev = ix->device;
        sh.lo = std::min(nq, (uint32_t)r * per);
        sh.n = std::min(nq, sh.lo + per) - sh.lo;
        if (sh.n == 0) continue;
        RG_HIP(hipSetDevice(ix->device));
        RG_HIP(hipStreamCreate(&sh.s));
        RG_HIP(hipMalloc(&sh.q, (size_t)sh.n * d * 4));
        RG_HIP(hipMalloc(&sh.ids, (size_t)sh.n * k * 4));
        RG_HIP(hipMalloc(&sh.dist, (size_t)sh.n * k * 4));
        RG_HIP(hipMalloc(&sh.ch, (size_t)sh.n * 2 * 4));
        RG_HIP(hipMemcpyAsync(sh.q, hq.data() + (size_t)sh.lo * d, (size_t)sh.n * d * 4, hipMemcpyHostToDevice, sh.s));
        RG_HIP(hipMemsetAsync(sh.ids, 0, (size_t)sh.n * k * 4, sh.s));
        RG_HIP(hipMemsetAsync(sh.dist, 0, (size_t)sh.n * k * 4, sh.s));
        st = rg::search_dev(ix, sh.q, sh.n, d, k, L_pq, sh.ids, sh.dist, sh.ch, sh.ch + sh.n, sh.s);
    }
    // every launched slice is waited for, whatever happened to the others; the first failure (in query order) is reported
    rg_status first = st;
    std::string first_msg = st != RG_OK ? rg_last_error() : "";
    for (int r = 0; r < nrep; ++r) {
        Shard &sh = S[(size_t)r];
        if (!sh.s || sh.n == 0) continue;
        (void)hipSetDevice(sh.dev);
        rg_status w = rg::search_wait(replicas[r], sh.s, k);
        if (w == RG_OK || w == RG_ERR_NOT_ENOUGH) {
            (void)hipMemcpy(out_ids + (size_t)sh.lo * k, sh.ids, (size_t)sh.n * k * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(out_dists + (size_t)sh.lo * k, sh.dist, (size_t)sh.n * k * 4, hipMemcpyDeviceToHost);
            if (out_cmps) (void)hipMemcpy(out_cmps + sh.lo, sh.ch, (size_t)sh.n * 4, hipMemcpyDeviceToHost);
            if (out_hops) (void)hipMemcpy(out_hops + sh.lo, sh.ch + sh.n, (size_t)sh.n * 4, hipMemcpyDeviceToHost);
        }
        if (w != RG_OK && first == RG_OK) { first = w; first_msg = rg_last_error(); }
    }
    if (first != RG_OK) return set_error(first, first_msg);
    return RG_OK;
}

rg_status rg_score_batch_dev(rg_index *ix, const float *d_query, const uint32_t *d_ids, uint32_t n, float *d_out,
                             void *stream) {
    return rg::score_dev(ix, d_query, d_ids, n, d_out, (hipStream_t)stream);
}

rg_status rg_score_batch(rg_index *ix, const float *query, const uint32_t *ids, uint32_t n, float *out) {
    if (!ix || !query || !ids || !out) return set_error(RG_ERR_ARG, "null argument");
    if (n == 0) return RG_OK;
    RG_HIP(hipSetDevice(ix->device));
    for (uint32_t i = 0; i < n; ++i)
        if (ids[i] >= ix->nd) return set_error(RG_ERR_ARG, "id out of range");
    rg::DevBuf<float> d_q, d_o;
    rg::DevBuf<uint32_t> d_i;
    RG_HIP(d_q.alloc(ix->dim));
    RG_HIP(d_o.alloc(n));
    RG_HIP(d_i.alloc(n));
    RG_HIP(hipMemcpy(d_q.p, query, (size_t)ix->dim * 4, hipMemcpyHostToDevice));
    RG_HIP(hipMemcpy(d_i.p, ids, (size_t)n * 4, hipMemcpyHostToDevice));
    rg_status st = rg::score_dev(ix, d_q.p, d_i.p, n, d_o.p, nullptr);
    if (st == RG_OK) {
        hipError_t e = hipMemcpy(out, d_o.p, (size_t)n * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = set_error(RG_ERR_DEVICE, hipGetErrorString(e));
    }
    return st;
}

}  // extern "C"
