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
#define RG_PROF_DECL unsigned long long pf_t = clock64(), pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_m[4] = {0, 0, 0, 0}
#define RG_PROF_MERGE , pf_m
#define RG_PROF_MERGE_ARG , unsigned long long (&pf_m)[4]
#define RG_PROF_M(i) { const unsigned long long t_ = clock64(); pf_m[i] += t_ - pf_tm; pf_tm = t_; }
#define RG_PROF_M0 unsigned long long pf_tm = clock64();
#define RG_PROF(i) { const unsigned long long t_ = clock64(); pf_acc[i] += t_ - pf_t; pf_t = t_; }
#define RG_PROF_CNT(i, v) { pf_cnt[i] += (v); }
#else
#define RG_PROF_DECL
#define RG_PROF_MERGE
#define RG_PROF_MERGE_ARG
#define RG_PROF_M(i)
#define RG_PROF_M0
#define RG_PROF(i)
#define RG_PROF_CNT(i, v)
#endif

constexpr int kCand = 128;  // candidate ids / scores of one iteration held in LDS: up to 64 of the popped node + up to 63 speculated

struct Beam {
    uint2 *ent;  // LDS: x = distance bits, y = id | kFlagBit
    uint32_t size, cur, cap;
};

// ordering point for LDS traffic inside a single-wave workgroup: the LDS unit executes one wave's DS instructions in
// issue order, so a write by one lane is seen by a later read of any lane without waiting; only the compiler has to be
// kept from moving accesses across
__device__ __forceinline__ void lds_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

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
    lds_fence();
    return make_uint2(e.x, e.y & ~kFlagBit);
}

// Insert the scored candidates (one per lane with have == true) -- the net effect of that many calls of
// NeighborPriorityQueue::insert (neighbor.h:150-183).  The beam is the top-cap of everything inserted so far under the
// total order (distance, id); candidates are distinct unvisited nodes, the only possible repeat is the entry point (never
// marked visited, index_bipartite.cpp:2349), whose second insert the reference drops.
//
// The merge runs in two steps so that the caller can act between them:
//   merge_rank   tail test, rank of every candidate among the beam entries and among each other, and from those the
//                position of the cursor after the merge -- i.e. WHICH ENTRY IS POPPED NEXT is known here, before a single
//                entry has moved (the hop loop requests that node's adjacency row at this point, so that the row's latency
//                runs under the shifting)
//   merge_apply  shifts the beam entries and writes the candidates in
//   mscr   128 words of LDS scratch
//   track  in: beam index of an entry to follow through the merge (or ~0u); out: its index afterwards, ~0u if it fell off
struct MergePlan {
    bool any;            // a candidate enters the beam
    bool valid, keep;    // this lane's candidate is inserted / lands inside the capacity
    uint32_t nc, fpos, slo, minq, ncur;
    uint32_t s0, s1, s2, s3;   // insertion ranks of the first four candidates (wave-uniform; ~0u beyond nc)
};

template <bool DEDUP>
__device__ __forceinline__ MergePlan merge_rank(const Beam &bm, float cd, uint32_t cid, bool have, uint32_t ep, int lane, uint32_t *mscr,
                                                uint32_t &track RG_PROF_MERGE_ARG) {
    RG_PROF_M0;
    MergePlan mp;
    mp.any = false; mp.valid = mp.keep = false; mp.nc = 0; mp.fpos = mp.slo = mp.minq = 0xffffffffu;
    mp.s0 = mp.s1 = mp.s2 = mp.s3 = 0xffffffffu;
    mp.ncur = bm.cur < bm.size ? bm.cur : 0xffffffffu;   // nothing enters: the cursor stays where it is
    bool valid = have && cid != ep;
    if (bm.size == bm.cap) {  // full: only candidates better than the current worst can enter (neighbor.h:151-153)
        uint2 w = bm.ent[bm.cap - 1];
        valid = valid && nb_less(cd, cid, __uint_as_float(w.x), w.y & ~kFlagBit);
    }
    if (!__any(valid)) { RG_PROF_M(0); return mp; }
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
        // A node can reach the merge a second time (forgetful visited filter, or a speculated list that turns out to
        // hold a visited node).  Its distance bits are the same, so the lower bound lands exactly on its beam entry if it
        // is still there: drop it (the reference's equal-id rule, neighbor.h:161); if it was evicted or never got in the
        // tail test above already rejected it (the tail only improves).
        if (valid && lo < bm.size && (bm.ent[lo].y & ~kFlagBit) == cid) valid = false;
        // The same id twice in one merge: one of them stays (they carry the same distance bits, so it does not matter
        // which).  Election through a 64-slot table keyed by the id: every candidate writes its lane to its slot, the
        // survivor of the slot keeps its candidate, a loser with the survivor's id is the duplicate and drops out, a loser
        // with another id (slot collision) is settled by the exact pairwise loop below -- rare.
        const uint32_t slot = (cid * 0x9E3779B1u) >> 26;
        if (valid) mscr[slot] = (uint32_t)lane;
        lds_fence();
        const uint32_t wl = valid ? mscr[slot] : (uint32_t)lane;
        lds_fence();
        const uint32_t wid = (uint32_t)__shfl((int)cid, (int)wl, 64);     // the slot survivor's id
        const bool won = valid && wl == (uint32_t)lane;
        if (valid && !won && wid == cid) valid = false;
        // undecided: lost the slot to ANOTHER id.  A twin of such a lane lost the same slot the same way, so twins are
        // looked for among the undecided lanes only; the lowest lane of a set of twins stays
        const bool unsure = valid && !won;
        const unsigned long long mu = __ballot(unsure);
        bool dup = false;
        for (unsigned long long m = mu; m; m &= m - 1) {
            const int sl = __ffsll((long long)m) - 1;
            dup = dup || (readlane_u(cid, sl) == cid && sl < lane);
        }
        valid = valid && !(unsure && dup);
    }
    const unsigned long long vmask = __ballot(valid);
    RG_PROF_M(0);
    if (!vmask) return mp;
    const uint32_t nc = __popcll(vmask);
    // rank among the candidates
    uint32_t crank = 0;
    for (unsigned long long m = vmask; m; m &= m - 1) {
        const int s = __ffsll((long long)m) - 1;
        const float od = readlane_f(cd, s);
        const uint32_t oi = readlane_u(cid, s);
        crank += nb_less(od, oi, cd, cid) ? 1u : 0u;
    }
    mp.any = true; mp.valid = valid; mp.nc = nc;
    mp.fpos = lo + crank;
    mp.keep = valid && mp.fpos < bm.cap;
    // insertion ranks in candidate order: lane j <- the beam rank of the candidate of rank j (non-decreasing in j)
    if (valid) mscr[crank] = lo;
    lds_fence();
    mp.slo = (uint32_t)lane < nc ? mscr[lane] : 0xffffffffu;
    lds_fence();
    mp.s0 = readlane_u(mp.slo, 0); mp.s1 = readlane_u(mp.slo, 1); mp.s2 = readlane_u(mp.slo, 2); mp.s3 = readlane_u(mp.slo, 3);
    mp.minq = mp.s0;                                    // first beam index that moves
    // new cursor: first unflagged entry after the merge
    uint32_t ncur = 0xffffffffu;
    if (bm.cur < bm.size) {
        const uint32_t sh = __popcll(__ballot((uint32_t)lane < nc && mp.slo <= bm.cur));
        if (bm.cur + sh < bm.cap) ncur = bm.cur + sh;
    }
    mp.ncur = min(ncur, wave_min_u32(mp.keep ? mp.fpos : 0xffffffffu));
    if (track != 0xffffffffu) {
        const uint32_t sh = __popcll(__ballot((uint32_t)lane < nc && mp.slo <= track));
        track = track + sh < bm.cap ? track + sh : 0xffffffffu;
    }
    RG_PROF_M(1);
    return mp;
}

// (distance bits, id) of the entry the cursor will point at once the plan is applied: an old entry that only shifts, or
// one of the candidates; valid when mp.ncur != ~0u.  Wave-uniform.
__device__ __forceinline__ uint2 merge_next_entry(const Beam &bm, const MergePlan &mp, float cd, uint32_t cid, int lane) {
    const unsigned long long from_cand = __ballot(mp.keep && mp.fpos == mp.ncur);
    if (from_cand) {
        const int s = __ffsll((long long)from_cand) - 1;
        return make_uint2(__float_as_uint(readlane_f(cd, s)), readlane_u(cid, s));
    }
    const uint2 e = bm.ent[bm.cur];     // the old cursor entry, moved right by the candidates ranked at or before it
    return make_uint2(e.x, e.y & ~kFlagBit);
}

__device__ __forceinline__ void merge_apply(Beam &bm, const MergePlan &mp, float cd, uint32_t cid, int lane RG_PROF_MERGE_ARG) {
    if (!mp.any) return;
    RG_PROF_M0;
#ifdef RG_K1_PROF
    pf_m[3] += ((unsigned long long)mp.nc << 32) | ((bm.size - mp.minq + 63) >> 6);
#endif
    // shift entries [minq, size) right by the number of candidates ranked at or before them.  Done in place, top group
    // first; a group is up to G chunks of 64 entries held in registers, so its reads all complete before its writes
    // (which only land on indices >= the ones read, i.e. inside the group or in groups already moved).  Hardly ever more
    // than four candidates enter per hop (2.5 - 3 on a genuine index once the beam is full): their ranks sit in scalar
    // registers and an entry's shift is four compares; further candidates are walked one by one.
    constexpr int G = 4;
    const int minq = (int)mp.minq;
    for (int top = (int)bm.size - 1; top >= minq; top -= kWave * G) {
        uint2 e[G];
        uint32_t sh[G];
#pragma unroll
        for (int g2 = 0; g2 < G; ++g2) {
            const int i = top - kWave * g2 - lane;
            e[g2] = i >= minq ? bm.ent[i] : make_uint2(0, 0);
            const uint32_t ui = (uint32_t)max(i, 0);
            sh[g2] = (mp.s0 <= ui ? 1u : 0u) + (mp.s1 <= ui ? 1u : 0u) + (mp.s2 <= ui ? 1u : 0u) + (mp.s3 <= ui ? 1u : 0u);
        }
        for (uint32_t j = 4; j < mp.nc; ++j) {
            const uint32_t l = readlane_u(mp.slo, (int)j);
#pragma unroll
            for (int g2 = 0; g2 < G; ++g2) sh[g2] += (int)l <= top - kWave * g2 - lane ? 1u : 0u;
        }
        lds_fence();
#pragma unroll
        for (int g2 = 0; g2 < G; ++g2) {
            const int i = top - kWave * g2 - lane;
            if (i >= minq && (uint32_t)i + sh[g2] < bm.cap) bm.ent[(uint32_t)i + sh[g2]] = e[g2];
        }
        lds_fence();
    }
    if (mp.keep) bm.ent[mp.fpos] = make_uint2(__float_as_uint(cd), cid);
    bm.size = min(bm.cap, bm.size + mp.nc);
    bm.cur = mp.ncur == 0xffffffffu ? bm.size : mp.ncur;
    lds_fence();
    RG_PROF_M(2);
}

template <bool DEDUP>
__device__ __forceinline__ void beam_merge(Beam &bm, float cd, uint32_t cid, bool have, uint32_t ep, int lane, uint32_t *mscr,
                                           uint32_t &track RG_PROF_MERGE_ARG) {
    const MergePlan mp = merge_rank<DEDUP>(bm, cd, cid, have, ep, lane, mscr, track RG_PROF_MERGE);
    merge_apply(bm, mp, cd, cid, lane RG_PROF_MERGE);
}

// DIMC: 0 = any dimension (query staged in LDS), else the compile-time dimension (query in registers)
// BF:   opt-in fast mode, NOT parity: the traversal scores a bf16 copy of the base (4 instead of 7 HBM
//       lines per d = 200 evaluation); at the end the whole beam is re-scored with the exact fp32 routine and the k best
//       by exact (distance, id) are returned, so the reported distances are exact for the returned ids.
//
// Speculative second expansion (P.spec, ELL adjacency): together with the popped node's adjacency row the kernel fetches
// the row of the entry that is NEXT in line (the closest unexpanded entry after the pop), screens its neighbours against
// the visited set WITHOUT marking them, and gathers + scores them in the same passes as the popped node's.  After the
// popped node's candidates are merged, the next pop is known: if it is the speculated node (about 90 % of the hops on a
// genuine index at L_pq >= 500) its candidates are marked visited, counted and merged right away -- two hops on one
// adjacency / visited / gather latency chain.  If not, the speculated scores are dropped and nothing else happened: no
// visited mark, no count, no beam change.  Results are therefore bit-identical with and without speculation; a miss
// costs the row reads of the dropped candidates.  P.spec == 2 (opt-in, NOT parity) expands the speculated node even on
// a miss, as long as it is still in the beam.
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
    uint32_t *cand_id = reinterpret_cast<uint32_t *>(qv + (DIMC ? 0u : P.dim));   // kCand
    float qr[DIMC ? (DIMC + 15) / 16 : 1];
    float qb[BF ? 8 * NB : 1];
    float *cand_d = reinterpret_cast<float *>(cand_id + kCand);           // kCand
    uint32_t *mscr = reinterpret_cast<uint32_t *>(cand_d + kCand);        // 128: merge scratch
    Beam bm;
    uint32_t *adjbuf = mscr + 2 * kWave;                                  // 64: prefetched adjacency row (LDS-DMA)
    bm.ent = reinterpret_cast<uint2 *>(adjbuf + kWave);                   // L
    bm.cap = P.L;
    // VIS=1: lossy exact-match visited filter (direct mapped, 16-bit remainders of a bijective id hash)
    // id-log staging: ids are appended here and flushed to HBM 64 at a time (256-B aligned full-line stores; small
    // unaligned appends would turn into read-modify-writes at the memory side once the line has left L2)
    uint32_t *logbuf = reinterpret_cast<uint32_t *>(bm.ent + P.L);        // 128
    uint16_t *vtab = reinterpret_cast<uint16_t *>(logbuf + 128);
    const uint32_t vf_rem_bits = P.id_bits > P.vf_slots_log2 ? P.id_bits - P.vf_slots_log2 : 0u;
    const uint32_t vf_id_mask = P.id_bits >= 32u ? 0xffffffffu : ((1u << P.id_bits) - 1u);
    const uint32_t vf_rem_mask = (1u << vf_rem_bits) - 1u;

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
        // LDS filter slot / remainder of an id (bijective hash: odd multiplier mod 2^id_bits)
        auto vf_hash = [&](uint32_t id, uint32_t &slot, uint16_t &rem) __attribute__((always_inline)) {
            const uint32_t x = (id * 0x9E3779B1u) & vf_id_mask;
            slot = x >> vf_rem_bits;
            rem = (uint16_t)(x & vf_rem_mask);
        };
        // visited test-and-set of this lane's neighbour (:2378, :2385); same-hop duplicates are resolved by the atomic's
        // order (VIS = 0) or by the merge's de-duplication (VIS = 1)
        auto visit_set = [&](uint32_t id, bool have) __attribute__((always_inline)) -> bool {
            bool fresh = false;
            if (VIS == 1) {
                // exact-match lookup: a hit proves "visited"; a miss is treated as fresh (may re-score a node whose
                // entry was overwritten -- harmless for the beam, see beam_merge<true>)
                if (have) {
                    uint32_t slot; uint16_t rem;
                    vf_hash(id, slot, rem);
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
                    uint32_t slot; uint16_t rem;
                    vf_hash(id, slot, rem);
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
            return fresh;
        };
        // the same test WITHOUT the set (speculated list).  A stale answer can only err towards "fresh": the list is
        // screened again, with the set, when (if) it is consumed.
        auto visit_peek = [&](uint32_t id, bool have) __attribute__((always_inline)) -> bool {
            if (!have) return false;
            if (VIS == 1 || P.vf_front) {
                uint32_t slot; uint16_t rem;
                vf_hash(id, slot, rem);
                if (vtab[slot] == rem) return false;
                if (VIS == 1) return true;
            }
            const uint32_t w = __hip_atomic_load(&vmap[id >> 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return !((w >> 16) == epoch && (w & (1u << (id & 15u))));
        };
        // id log (VIS = 1): n ids from cand_id[off ..) join the LDS line buffer; a full 64-id line leaves as one aligned
        // 256-B store.  The store is issued where no gather is outstanding: in front of gathers it would sit at the head
        // of the vmcnt queue and put its completion latency on the critical path of the first counted wait.
        auto log_append = [&](uint32_t off, uint32_t n) __attribute__((always_inline)) {
            if (VIS == 1 && qlog) {
                if ((uint32_t)lane < n) logbuf[lbn + lane] = cand_id[off + lane];
                lbn += n;
            }
            logn += n;
        };
        auto log_flush = [&]() __attribute__((always_inline)) {
            if (VIS == 1 && qlog && lbn >= (uint32_t)kWave) {
                lds_fence();
                const uint32_t pos = logn - lbn;                       // ids already flushed (multiple of 64)
                const uint32_t v = logbuf[lane], rest = logbuf[kWave + lane];
                lds_fence();
                lbn -= kWave;
                if ((uint32_t)lane < lbn) logbuf[lane] = rest;
                if (pos + lane < P.logcap) qlog[pos + lane] = v;
                lds_fence();
            }
        };
        // gather + score (:2387) of cand_id[0 .. n) into cand_d, 4 rows per pass.
        //  * compile-time dimension (the BASELINE shapes): REGISTER-STAGED.  A batch of R passes (4R rows) is fetched with
        //    plain global_load_dwordx4 into R register sets -- all of them in flight together, one HBM latency per batch --
        //    and each set is then bounced through ONE LDS buffer in the slot layout the scoring routine reads (the layout
        //    the LDS-DMA path produces: lane l's 16 bytes at 16 l inside each 1-KiB block).  Rows in flight cost VGPRs, of
        //    which a latency-bound wave has plenty, instead of LDS, which is what limits resident queries at large L_pq.
        //  * otherwise (any dimension, or the bf16 fast mode): LDS-DMA into a ring of R staging buffers; pass p is consumed
        //    once only the loads of the passes issued after it are still outstanding.
        auto gather_list = [&](uint32_t n) __attribute__((always_inline)) {
            const uint32_t npass = (n + 3u) >> 2;
            if constexpr (DIMC != 0 && !BF) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                constexpr int NBLK = (DIMC + 63) / 64, NFULL = DIMC / 64, REM = DIMC & 63;
                const int jsrc = ((lane & 15) - 4 * g) & 15;
                const bool tail = 4 * jsrc < REM;
                const uint32_t rid0 = cand_id[0];
                for (uint32_t p0 = 0; p0 < npass; p0 += R) {
                    v4f rv[R][NBLK];
                    uint32_t rid[R];
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        const uint32_t c = 4 * (p0 + j) + g;
                        // idle groups and idle sets re-read a row that is being fetched anyway: every set is loaded
                        // unconditionally so that the waits in front of the sets are exact counts (a set behind a branch
                        // makes the compiler wait for ALL sets before the first one is scored: -10 % measured)
                        rid[j] = c < n ? cand_id[c] : rid0;
                    }
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        const float *src = P.base + (size_t)rid[j] * P.stride + 4 * jsrc;
#pragma unroll
                        for (int b = 0; b < NFULL; ++b) rv[j][b] = *reinterpret_cast<const v4f *>(src + 64 * b);
                        if constexpr (REM != 0) {
                            v4f t = {0.0f, 0.0f, 0.0f, 0.0f};
                            if (tail) t = *reinterpret_cast<const v4f *>(src + 64 * NFULL);
                            rv[j][NFULL] = t;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        if (p0 + j < npass) {
#pragma unroll
                            for (int b = 0; b < NBLK; ++b) *reinterpret_cast<v4f *>(stage + 256 * b + 4 * lane) = rv[j][b];
                            RG_PROF(6);
                            lds_fence();
                            const float d = score(stage);
                            const uint32_t c = 4 * (p0 + j) + g;
                            if (c < n && (lane & 15) == 0) cand_d[c] = d;
                            lds_fence();
                            RG_PROF(3);
                        }
                    }
                }
            } else {
                const uint32_t lpp = BF ? (uint32_t)NB : loads_per_pass(P.dim);
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
                    RG_PROF(3);
                }
            }
        };

        // entry point: scored and queued, not marked visited (index_bipartite.cpp:2338-2352)
        issue(P.ep, g == 0, stage);
        gather_wait(0);
        const float epd = score(stage);
        if (lane == 0) bm.ent[0] = make_uint2(__float_as_uint(epd), P.ep);
        bm.size = 1;
        bm.cur = 0;
        wave_sync();

        uint32_t cmps = 0, hops = 0;
        // adjacency row of the node that will be popped next, requested as soon as the merge knows which one it is
        uint32_t pre_node = 0xffffffffu;
        RG_PROF(5);
        while (bm.cur < bm.size) {                                         // has_unexpanded_node, :2356
            const uint2 popped = beam_pop(bm, lane);                       // :2358
            const uint32_t node = popped.y;
            if (build && lane == 0 && hops < P.exp_cap) P.out_exp[(size_t)qi * P.exp_cap + hops] = popped;   // full_retset, :1319
            ++hops;                                                        // :2366
            RG_PROF(0);
            // adjacency of `node`, 64 words at a time; with it (ELL) the row of the entry that is now next in line
            uint32_t deg, first = 0, first2 = 0, node2 = 0xffffffffu, pos2 = 0xffffffffu;
            const uint32_t *list;
            if (ELL) {
                const uint32_t *row = P.ell + (size_t)node * P.ell_stride;
                if (pre_node == node) {                                    // requested during the previous hop's merge
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    first = (uint32_t)lane < P.ell_stride ? adjbuf[lane] : 0u;
                } else {
                    first = (uint32_t)lane < P.ell_stride ? row[lane] : 0u;
                }
                pre_node = 0xffffffffu;
                if (P.spec && !P.diag && bm.cur < bm.size) {
                    pos2 = bm.cur;
                    node2 = bm.ent[pos2].y;                                // unflagged by the cursor invariant
                    const uint32_t *row2 = P.ell + (size_t)node2 * P.ell_stride;
                    first2 = (uint32_t)lane < P.ell_stride ? row2[lane] : 0u;
                }
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
            RG_PROF_CNT(5, deg);
            uint32_t deg2 = 0;
            if (ELL && node2 != 0xffffffffu) {
                deg2 = readlane_u(first2, 0);
                if (deg > 63u || deg2 > 63u || deg2 == 0u) node2 = 0xffffffffu;   // wide or empty rows take the plain path
            }
            if (ELL && node2 != 0xffffffffu) {
                // ---- pair path: candidates of `node` (A) and, speculatively, of `node2` (B) share one gather phase
                uint32_t idA = (uint32_t)__shfl_down((int)first, 1, 64), idB = (uint32_t)__shfl_down((int)first2, 1, 64);
                bool haveA = (uint32_t)lane < deg, haveB = (uint32_t)lane < deg2;
                if (build) { haveA = haveA && idA != tgt; haveB = haveB && idB != tgt; }
                const bool freshA = visit_set(idA, haveA);
                const bool freshB = visit_peek(idB, haveB);      // after A's marks: common neighbours are not gathered twice
                const unsigned long long fmA = __ballot(freshA), fmB = __ballot(freshB);
                const uint32_t nA = __popcll(fmA), nB = __popcll(fmB);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (freshA) cand_id[__popcll(fmA & below)] = idA;
                if (freshB) cand_id[nA + __popcll(fmB & below)] = idB;
                lds_fence();
                log_append(0, nA);
                cmps += nA;                                                // :2397
                RG_PROF_CNT(0, 1); RG_PROF_CNT(1, nA); RG_PROF_CNT(2, 1);
                RG_PROF(2);
                if (nA + nB) gather_list(nA + nB);
                log_flush();
                // queue inserts of A (:2398)
                {
                    const float cd = (uint32_t)lane < nA ? cand_d[lane] : 0.0f;
                    const uint32_t cid = (uint32_t)lane < nA ? cand_id[lane] : 0u;
                    lds_fence();
                    RG_PROF(3);
                    if (nA) beam_merge<VIS == 1>(bm, cd, cid, (uint32_t)lane < nA, P.ep, lane, mscr, pos2 RG_PROF_MERGE);
                    RG_PROF(4);
                }
                // the next pop is known now: is it the speculated node?
                const bool hit = bm.cur < bm.size && pos2 == bm.cur;
                RG_PROF_CNT(3, hit ? 1 : 0);
                if (hit || (P.spec == 2u && pos2 != 0xffffffffu)) {
                    uint2 e2;
                    if (hit) e2 = beam_pop(bm, lane);                      // :2358 of the second hop
                    else {                                                 // multi_expand, miss: expanded out of order (NOT parity)
                        e2 = bm.ent[pos2];
                        if (lane == 0) bm.ent[pos2].y = e2.y | kFlagBit;
                        lds_fence();
                    }
                    if (build && lane == 0 && hops < P.exp_cap) P.out_exp[(size_t)qi * P.exp_cap + hops] = make_uint2(e2.x, e2.y & ~kFlagBit);
                    ++hops;
                    RG_PROF(0);
                    // B becomes the expansion of node2: mark, count, log, merge
                    const float cd = (uint32_t)lane < nB ? cand_d[nA + lane] : 0.0f;
                    const uint32_t cid = (uint32_t)lane < nB ? cand_id[nA + lane] : 0u;
                    bool validB = (uint32_t)lane < nB;
                    uint32_t cntB = nB;
                    if (VIS == 1) {
                        if (validB) { uint32_t slot; uint16_t rem; vf_hash(cid, slot, rem); vtab[slot] = rem; }
                        log_append(nA, nB);
                    } else {
                        // authoritative test-and-set; a candidate that turns out visited (stale peek, repeated edge) is
                        // not counted, and the merge's de-duplication drops it
                        const bool fr = visit_set(cid, validB);
                        cntB = __popcll(__ballot(fr));
                        validB = fr;
                    }
                    cmps += cntB;
                    RG_PROF_CNT(0, 1); RG_PROF_CNT(1, cntB);
                    lds_fence();
                    log_flush();
                    RG_PROF(2);
                    uint32_t none = 0xffffffffu;
                    if (nB) beam_merge<true>(bm, cd, cid, validB, P.ep, lane, mscr, none RG_PROF_MERGE);
                    RG_PROF(4);
                }
                continue;
            }
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
                const bool fresh = visit_set(id, have);
                const unsigned long long fm = __ballot(fresh);
                const uint32_t n = __popcll(fm);
                RG_PROF_CNT(0, 1); RG_PROF_CNT(1, n);
                if (n == 0) { RG_PROF(2); continue; }
                if (fresh) cand_id[__popcll(fm & ((1ull << lane) - 1ull))] = id;
                lds_fence();
                log_append(0, n);
                cmps += n;                                                 // :2397
                RG_PROF(2);
                gather_list(n);
                log_flush();
                // queue inserts (:2398)
                const float cd = (uint32_t)lane < n ? cand_d[lane] : 0.0f;
                const uint32_t cid = (uint32_t)lane < n ? cand_id[lane] : 0u;
                lds_fence();
                RG_PROF(3);
                uint32_t none = 0xffffffffu;
                if (ELL && c0 + kWave >= deg && !(P.diag & 4u)) {
                    // last chunk of the hop: once the ranks are known the next pop is known -- request its adjacency row
                    // now, its latency runs under the shifting and the pop
                    const MergePlan mp = merge_rank<VIS == 1>(bm, cd, cid, (uint32_t)lane < n, P.ep, lane, mscr, none RG_PROF_MERGE);
                    if (mp.ncur != 0xffffffffu) {
                        // LDS-DMA through inline asm: the row has no register destination the compiler could want to wait
                        // for (it parked an s_waitcnt vmcnt(0) at the top of the shift loop when this was a plain load);
                        // it is waited for explicitly where the next hop picks it up
                        pre_node = merge_next_entry(bm, mp, cd, cid, lane).y;
                        const uint32_t *src = P.ell + (size_t)pre_node * P.ell_stride + lane;
                        const uint32_t lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t *)adjbuf);
                        if ((uint32_t)lane < P.ell_stride)
                            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(lds_addr), "v"(src) : "memory");
                    }
                    merge_apply(bm, mp, cd, cid, lane RG_PROF_MERGE);
                } else {
                    beam_merge<VIS == 1>(bm, cd, cid, (uint32_t)lane < n, P.ep, lane, mscr, none RG_PROF_MERGE);
                }
                RG_PROF(4);
            }
        }

        // results (:2408-2418)
        wave_sync();
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
            for (int i = 0; i < 8; ++i) { P.prof[(size_t)qi * 24 + i] = pf_acc[i]; P.prof[(size_t)qi * 24 + 8 + i] = pf_cnt[i]; }
            for (int i = 0; i < 4; ++i) P.prof[(size_t)qi * 24 + 16 + i] = pf_m[i];
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
    int *occupancy = nullptr;   // non-null: do not launch, report the resident single-wave workgroups per CU of the kernel
};

template <bool L2, bool ELL, int R, int VIS, int DIMC, bool BF = false>
static rg_status launch_search_d(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    auto kern = rg_search_kernel<L2, ELL, R, VIS, DIMC, BF>;
    RG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds));
    if (c.occupancy) {   // registers as well as LDS bound the resident queries (the register-staged forms are VGPR-heavy)
        RG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(c.occupancy, reinterpret_cast<const void *>(kern), 64, c.lds));
        return RG_OK;
    }
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
    if constexpr (ELL) {   // eight register sets (32 rows in flight): d = 200 only, for launches with few resident queries
        if (c.R == 8 && c.dimc == 200 && !c.bf)
            return c.vis == 1 ? launch_search_d<L2, ELL, 8, 1, 200>(P, c, s) : launch_search_d<L2, ELL, 8, 0, 200>(P, c, s);
    }
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
