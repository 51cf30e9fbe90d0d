// rg_search_kernel.h -- K1, the persistent beam-search kernel (one wave64 per in-flight query), and its launch plumbing.
//
//   K1  rg_search_kernel  == IndexBipartite::SearchRoarGraph (src/index_bipartite.cpp:2311-2420)
//
// Per query (one wave, one single-wave workgroup, all state wave-private):
//   LDS   : sorted beam of L_pq (dist, id|expanded) pairs  == NeighborPriorityQueue (neighbor.h:138-223)
//           query vector, candidate id/score scratch, LDS-DMA staging for the row gather, visited filter, id-log line
//   HBM   : (visited mode 0) epoch-tagged visited words per slot == VisitedList (visited_list_pool.h:8-29)
// Per hop (hop-synchronous, see SURVEY.md Appendix C-11 for why this reproduces the sequential inserts):
//   pop closest unexpanded -> read its adjacency row -> visited test-and-set -> ballot-compact the unvisited ids
//   -> gather + score them 4 rows per sub-pass -> rank-merge the survivors into the beam.
//
// The template is instantiated in four translation units (rg_search_inst_*.hip: metric x adjacency layout) so that they
// compile in parallel; rg_search.hip holds the host side.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "rg.h"
#include "rg_device.h"
#include "rg_internal.h"

namespace rg {

#ifndef RG_HIP
#define RG_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return set_error(RG_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)
#endif

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
    uint32_t spec;            // 1 = speculative second expansion per hop (bit-exact), 2 = merged unconditionally (opt-in, NOT parity)
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

constexpr int kCand = 64;   // candidate ids / scores of one hop held in LDS

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
    float *cand_d = reinterpret_cast<float *>(cand_id + kCand);
    Beam bm;
    bm.ent = reinterpret_cast<uint2 *>(cand_d + kCand);                   // L
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
                        RG_PROF(6);
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


// ------------------------------------------------------------------------------------------ launch plumbing
// what the host decided for one launch (rg_search.hip: plan_k1)
struct K1Launch {
    int R = 1;        // staging ring depth (passes of 4 rows in flight)
    int vis = 1;      // 0 = exact HBM visited words, 1 = LDS filter
    int dimc = 0;     // compile-time dimension instantiation (0 = generic)
    bool bf = false;  // opt-in bf16 traversal
    uint32_t grid = 0;
    size_t lds = 0;
};

template <bool L2, bool ELL, int R, int VIS, int DIMC, bool BF = false>
static rg_status launch_search_d(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    auto kern = rg_search_kernel<L2, ELL, R, VIS, DIMC, BF>;
    RG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds));
    hipLaunchKernelGGL(kern, dim3(c.grid), dim3(64), c.lds, s, P);
    RG_HIP(hipGetLastError());
    return RG_OK;
}

template <bool L2, bool ELL, int R, int VIS>
static rg_status launch_search_v(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    if constexpr (ELL) {
        if constexpr (R <= 2) {
            if (c.bf && c.dimc == 200) return launch_search_d<L2, ELL, R, VIS, 200, true>(P, c, s);
            if (c.bf && c.dimc == 512) return launch_search_d<L2, ELL, R, VIS, 512, true>(P, c, s);
        }
        if (c.dimc == 200) return launch_search_d<L2, ELL, R, VIS, 200>(P, c, s);
        if (c.dimc == 512) return launch_search_d<L2, ELL, R, VIS, 512>(P, c, s);
    }
    return launch_search_d<L2, ELL, R, VIS, 0>(P, c, s);
}

template <bool L2, bool ELL, int R>
static rg_status launch_search_t(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    return c.vis == 1 ? launch_search_v<L2, ELL, R, 1>(P, c, s) : launch_search_v<L2, ELL, R, 0>(P, c, s);
}

// every instantiation of one (metric, adjacency layout) family; one translation unit each (rg_search_inst_*.hip)
template <bool L2, bool ELL>
static rg_status launch_search_family(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    switch (c.R) {
        case 1: return launch_search_t<L2, ELL, 1>(P, c, s);
        case 2: return launch_search_t<L2, ELL, 2>(P, c, s);
        default: return launch_search_t<L2, ELL, 4>(P, c, s);
    }
}

rg_status launch_search_ip_ell(const SearchParams &P, const K1Launch &c, hipStream_t s);
rg_status launch_search_l2_ell(const SearchParams &P, const K1Launch &c, hipStream_t s);
rg_status launch_search_ip_csr(const SearchParams &P, const K1Launch &c, hipStream_t s);
rg_status launch_search_l2_csr(const SearchParams &P, const K1Launch &c, hipStream_t s);

}  // namespace rg
