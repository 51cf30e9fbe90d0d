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
#include <type_traits>

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
    uint32_t vf_slots;        // entries (16-bit) of the LDS visited filter: any multiple of 8 (powers of two included)
    uint32_t vf_rem_bits;     // bits of an entry: the smallest r with 2^r >= ceil(2^id_bits / vf_slots), at most 15
    uint32_t vf_front;        // VIS=0: 1 = the LDS filter screens the exact HBM words; VIS=2: 1 = the region is the bit screen
    uint32_t roll;            // register-staged gather: 1 = streamed (set j re-loaded as soon as it is scored), 0 = batch by batch
    uint32_t ls_front;        // VIS=2 with byte tags: 1 = the front of the LDS region is an exact set (vf_slots entries + vs_side words, as in VIS=3), the screen is the bl_words behind it
    uint32_t bl_words;        // words of that screen
    uint32_t vbytes;          // VIS=2: 1 = `visited` holds one epoch BYTE per node ([slots][4 * vwords] bytes) instead of the words
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
    // split rows (d = 200, ELL): `base` is then a copy of the first 192 elements of every row at a 768-B stride (six
    // whole 128-B lines), and the 8-element tails live once per EDGE, in adjacency order: the tails of one node's
    // neighbours are contiguous (tail_off[node] + position in the row), so a hop reads them from ceil(deg/4) lines
    // instead of a seventh line per fresh row.  Not split: tail_base = base + 192, tail_stride = stride, index = row id.
    const float *tail_base;
    uint32_t tail_stride;
    const uint32_t *tail_off; // [nd] first edge of the node (null = rows are not split)
    uint32_t ep_tail;         // tail slot of the entry point (it is scored without an edge leading to it)
    uint32_t spec;            // 2 = "multi_expand": two expansions per iteration (opt-in, NOT parity); 0 = the reference's order
    // ELL neighbour words of indexes with at most 2^24 nodes carry min(255, in-degree of the neighbour) in their top byte
    // (id_mask = 0x00ffffff; 0xffffffff = untagged).  A node is tested at most in-degree times per query, so the LDS
    // filter only spends entries on nodes whose in-degree reaches vf_min_indeg: the others are never remembered (and at
    // worst scored again the few times they are met again) -- same results, more of the filter for the nodes that return.
    uint32_t id_mask, vf_min_indeg;
    // in-kernel exact distinct count (round 3): narrow beams log a few thousand ids per query, which the wave that ran the
    // query can count itself when the query is over, in the LDS its beam and filter no longer need (K4's bucketed set,
    // count_tbits = log2 of its table words; 0 = off: K4 counts).  qlog_n[q] then carries kCountedBit and K4 skips the query.
    uint32_t count_tbits;
    // count_mode: 1 = the wave counts its query's log at the END OF THE QUERY (narrow beams of the filter + log form, round 3); 0 = K4 counts.
    // (Round 4 tried the counts in the TAIL of the launch, by the waves that found the work queue empty: exact, and slower -- dynamic
    // scheduling leaves no idle tail to hide them in)
    uint32_t count_mode;
    // VIS = 3, the EXACT set in LDS (round 4): the filter region holds vf_slots 16-bit entries in buckets of eight
    // (ds_read_b128 sees a bucket) + vs_side words of full ids for nodes whose bucket is full.  Nothing is ever evicted, so
    // no node is scored twice: cmps needs no log and no K4, the beam no de-duplication.  A query that outgrows the set goes
    // on in the forgetful form from that hop on (log + de-duplicating inserts) and counts its short log itself at the end;
    // lset_left counts such queries (the host stops using the form at a beam width where they are many).
    uint32_t vs_side;
    uint32_t *ovf_list, *ovf_count;     // VIS = 3: queries whose count could not be finished in the kernel (recounted by the host)
    unsigned long long *lset_left;
    unsigned long long *totals;   // [2] evaluations performed / distinct nodes of the queries counted here (as K4 reports them)
    // shared frontier (third mode; opt-in knob "shared_frontier"): every query of a batch starts at the entry
    // point, so the first expansion scores the same deg(ep) rows for all of them.  front_scores[q][0] = compare(ep, q) and
    // [q][1 + j] = compare(j-th neighbour of ep, q) were computed for the whole batch by rg_front_score_kernel (the exact
    // routine: same bits); the first hop reads them instead of gathering the rows.  null = off.
    const float *front_scores;
    uint32_t front_stride;
    // HUB BITS (round 5; VIS = 2): the first hub_words words of the visited region are an EXACT bitmap of
    // 2^hub_m bits for the "hubs" of that size -- per bit position p the node of highest in-degree among those with
    // (id * 0x9E3779B1) >> (32 - hub_m) == p (ties: the smaller id).  Positions nest (the top m - 1 bits of the product are the top
    // m bits shifted), so a node that owns its position at 2^m bits owns it at every larger size: the index stores, in the top
    // nibble of every ELL neighbour word, 8 less than the smallest m at which the neighbour is a hub (15 = never), and a launch with
    // 2^hub_m bits treats a neighbour as a hub iff 8 + nibble <= hub_m.  Two hubs never share a bit and no other node uses the
    // bitmap, so the bit IS the node's visited flag: no tag line read, no tag store, no set entry, no screen bit for it -- and a
    // third of all edges of the 10M bench index point into 1 % of its nodes.  0 = off.
    uint32_t hub_m, hub_words;
    uint32_t log_early;       // VIS = 1: the id-log store of a hop is issued right behind the row loads (else after the scoring)
    uint32_t look;            // VIS = 2: 1 = fetch the predicted next pop's adjacency row and visited words early, 0 = no speculation
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

constexpr int kCand = 128;  // candidate ids / scores of one iteration held in LDS (two expansions of up to 63 neighbours in the multi_expand mode)

// The beam == NeighborPriorityQueue (neighbor.h:138-223): the best `cap` of everything inserted so far under the total
// order (distance, id), each entry with an "expanded" flag, popped closest-unexpanded first.  Two containers hold it:
//   main     sorted array in LDS (x = distance bits, y = id | kFlagBit), `cur` = its first unflagged entry
//   pending  up to 64 recent insertions, sorted, ONE PER LANE in registers (lane j < psize holds the j-th)
// A hop inserts two or three entries on average (2.5 - 3 on a genuine index once the beam is full).  Into the sorted array
// each of them would shift everything behind it -- six 64-entry chunks per hop at L_pq = 2000, a third of the hop's time --
// so insertions go to the pending list (a few register operations) and the list is merged into the array in one pass when
// it is full (every twenty-odd hops).  Every query the search loop asks is answered over BOTH containers: the worst entry
// (tail test and eviction, neighbor.h:151-153, 168-170) is the worse of the two tails, the next pop (neighbor.h:185-192)
// the better of the two first unflagged entries -- so the sequence of pops and the final contents are those of the single
// sorted array.
struct Beam {
    uint2 *ent;
    uint32_t size, cur, cap;
    float pd;          // pending entry of this lane
    uint32_t pi;       // its id | kFlagBit
    uint32_t psize;
};

// ordering point for LDS traffic inside a single-wave workgroup: the LDS unit executes one wave's DS instructions in
// issue order, so a write by one lane is seen by a later read of any lane without waiting; only the compiler has to be
// kept from moving accesses across
__device__ __forceinline__ void lds_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool beam_has_unexpanded(const Beam &bm, int lane) {
    return bm.cur < bm.size || __ballot((uint32_t)lane < bm.psize && !(bm.pi & kFlagBit)) != 0ull;
}

// closest_unexpanded (neighbor.h:185-192) over both containers: flag the entry and return (distance bits, id)
__device__ __forceinline__ uint2 beam_pop(Beam &bm, int lane) {
    const unsigned long long pm = __ballot((uint32_t)lane < bm.psize && !(bm.pi & kFlagBit));
    const bool in_main = bm.cur < bm.size;
    uint2 e = make_uint2(0u, 0u);
    if (in_main) e = bm.ent[bm.cur];                         // unflagged by the cursor invariant
    if (pm) {
        const int j = __ffsll((long long)pm) - 1;            // sorted list: the first unflagged lane is the closest
        const float d = readlane_f(bm.pd, j);
        const uint32_t id = readlane_u(bm.pi, j);
        if (!in_main || nb_less(d, id, __uint_as_float(e.x), e.y)) {
            if (lane == j) bm.pi |= kFlagBit;
            return make_uint2(__float_as_uint(d), id);
        }
    }
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
    return e;
}

// the entry the next beam_pop would return if nothing were inserted before it (no state changes): the prediction of the
// look-ahead form (VIS = 2)
__device__ __forceinline__ bool beam_peek(const Beam &bm, int lane, float &d, uint32_t &id) {
    const unsigned long long pm = __ballot((uint32_t)lane < bm.psize && !(bm.pi & kFlagBit));
    const bool in_main = bm.cur < bm.size;
    if (!in_main && !pm) return false;
    uint2 e = make_uint2(0u, 0u);
    if (in_main) e = bm.ent[bm.cur];
    d = __uint_as_float(e.x);
    id = e.y;
    if (pm) {
        const int j = __ffsll((long long)pm) - 1;
        const float dj = readlane_f(bm.pd, j);
        const uint32_t ij = readlane_u(bm.pi, j);
        if (!in_main || nb_less(dj, ij, d, id)) { d = dj; id = ij; }
    }
    return true;
}

// rank of this lane's key among the main array's entries: lower bound under (distance, id)
__device__ __forceinline__ uint32_t beam_lower_bound(const Beam &bm, float cd, uint32_t cid, bool active) {
    uint32_t lo = 0, hi = active ? bm.size : 0;
    while (__any(lo < hi)) {
        if (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint2 e = bm.ent[mid];
            if (nb_less(__uint_as_float(e.x), e.y & ~kFlagBit, cd, cid)) lo = mid + 1;
            else hi = mid;
        }
    }
    return lo;
}

// merge the pending list into the main array: one pass, top chunk first.  An entry of the array moves right by the number
// of pending entries ranked at or before it; the pending entries carry their flags with them.
__device__ __forceinline__ void beam_flush(Beam &bm, int lane, uint32_t *mscr) {
    const uint32_t nc = bm.psize;
    if (nc == 0) return;
    const bool have = (uint32_t)lane < nc;
    const uint32_t cid = bm.pi & ~kFlagBit;
    const uint32_t lo = beam_lower_bound(bm, bm.pd, cid, have);       // rank among the array entries
    const uint32_t fpos = lo + (uint32_t)lane;                        // the list is sorted: lane j is its j-th entry
    // slo[j] = array rank of the j-th pending entry (non-decreasing in j): already one per lane
    const uint32_t slo = have ? lo : 0xffffffffu;
    const uint32_t minq = readlane_u(slo, 0);
    // cursor afterwards: the old one, moved by the pending entries ranked at or before it, or the first unflagged newcomer
    uint32_t ncur = 0xffffffffu;
    if (bm.cur < bm.size) ncur = bm.cur + (uint32_t)__popcll(__ballot(have && slo <= bm.cur));
    ncur = min(ncur, wave_min_u32(have && !(bm.pi & kFlagBit) ? fpos : 0xffffffffu));
    (void)mscr;
    constexpr int G = 4;
    int jh = (int)nc;
    for (int top = (int)bm.size - 1; top >= (int)minq; top -= kWave * G) {
        uint2 e[G];
        uint32_t sh[G];
#pragma unroll
        for (int g2 = 0; g2 < G; ++g2) {
            const int i = top - kWave * g2 - lane;
            e[g2] = i >= (int)minq ? bm.ent[i] : make_uint2(0, 0);
            sh[g2] = 0;
        }
        // shift of an entry = pending entries ranked at or before it.  Walk down the sorted ranks: those above the group's
        // top count for nobody (and for no later group either: jh only ever moves down), those at or below its bottom for
        // everybody, the ones in between are compared lane by lane
        const int gbot = top - kWave * G + 1;
        while (jh > 0 && (int)readlane_u(slo, jh - 1) > top) --jh;
        uint32_t all = 0;
        for (int j = jh - 1; j >= 0; --j) {
            const int l = (int)readlane_u(slo, j);
            if (l <= gbot) { all = (uint32_t)j + 1u; break; }
#pragma unroll
            for (int g2 = 0; g2 < G; ++g2) sh[g2] += l <= top - kWave * g2 - lane ? 1u : 0u;
        }
        lds_fence();
#pragma unroll
        for (int g2 = 0; g2 < G; ++g2) {
            const int i = top - kWave * g2 - lane;
            if (i >= (int)minq) bm.ent[(uint32_t)i + sh[g2] + all] = e[g2];
        }
        lds_fence();
    }
    if (have) bm.ent[fpos] = make_uint2(__float_as_uint(bm.pd), bm.pi);
    bm.size += nc;
    bm.cur = ncur == 0xffffffffu ? bm.size : ncur;
    bm.psize = 0;
    lds_fence();
}

// Insert the scored candidates (one per lane with have == true) -- the net effect of that many calls of
// NeighborPriorityQueue::insert (neighbor.h:150-183).  Candidates are distinct unvisited nodes, the only possible repeat
// is the entry point (never marked visited, index_bipartite.cpp:2349), whose second insert the reference drops.
//   mscr   128 words of LDS scratch
// DEDUP: a node can reach the beam a second time (forgetful visited filter).  Its distance bits are the same, so it is
// either still in the beam under exactly its key -- dropped like the reference's equal-id probe (neighbor.h:161) -- or it
// was evicted / never got in, and then the tail test rejects it (the tail only improves).
template <bool DEDUP>
__device__ __forceinline__ void beam_insert(Beam &bm, float cd, uint32_t cid, bool have, uint32_t ep, int lane, uint32_t *mscr RG_PROF_MERGE_ARG) {
    RG_PROF_M0;
    bool valid = have && cid != ep;
    if (bm.size + bm.psize == bm.cap) {   // full: only candidates better than the current worst can enter (neighbor.h:151-153)
        float wd = 0.0f;
        uint32_t wi = 0;
        bool w_main = bm.size > 0;
        if (w_main) { const uint2 w = bm.ent[bm.size - 1]; wd = __uint_as_float(w.x); wi = w.y & ~kFlagBit; }
        if (bm.psize > 0) {
            const float d2 = readlane_f(bm.pd, (int)bm.psize - 1);
            const uint32_t i2 = readlane_u(bm.pi, (int)bm.psize - 1) & ~kFlagBit;
            if (!w_main || nb_less(wd, wi, d2, i2)) { wd = d2; wi = i2; }
        }
        valid = valid && nb_less(cd, cid, wd, wi);
    }
    if (!__any(valid)) { RG_PROF_M(0); return; }
    if (DEDUP) {
        // still in the array?  the lower bound lands exactly on its entry
        const uint32_t lo = beam_lower_bound(bm, cd, cid, valid);
        if (valid && lo < bm.size && (bm.ent[lo].y & ~kFlagBit) == cid) valid = false;
        // still in the pending list?  (walk whichever side is shorter)
        const unsigned long long vm0 = __ballot(valid);
        if ((uint32_t)__popcll(vm0) <= bm.psize) {
            for (unsigned long long m = vm0; m; m &= m - 1) {
                const int s = __ffsll((long long)m) - 1;
                const uint32_t oi = readlane_u(cid, s);
                const bool hit = __ballot((uint32_t)lane < bm.psize && (bm.pi & ~kFlagBit) == oi) != 0ull;
                if (hit && lane == s) valid = false;
            }
        } else {
            for (uint32_t j = 0; j < bm.psize; ++j) valid = valid && (readlane_u(bm.pi, (int)j) & ~kFlagBit) != cid;
        }
        // The same id twice among the candidates: one of them stays (they carry the same distance bits, so it does not
        // matter which).  Election through a 64-slot table keyed by the id: every candidate writes its lane to its slot,
        // the survivor of the slot keeps its candidate, a loser with the survivor's id is the duplicate and drops out, a
        // loser with another id (slot collision) is undecided.  A twin of an undecided lane lost the same slot the same
        // way, so twins are looked for among the undecided lanes only; the lowest lane of a set of twins stays.
        const uint32_t slot = (cid * 0x9E3779B1u) >> 26;
        if (valid) mscr[slot] = (uint32_t)lane;
        lds_fence();
        const uint32_t wl = valid ? mscr[slot] : (uint32_t)lane;
        lds_fence();
        const uint32_t wid = (uint32_t)__shfl((int)cid, (int)wl, 64);     // the slot survivor's id
        const bool won = valid && wl == (uint32_t)lane;
        if (valid && !won && wid == cid) valid = false;
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
    if (!vmask) return;
    const uint32_t nc = __popcll(vmask);
    if (bm.psize + nc > (uint32_t)kWave) { beam_flush(bm, lane, mscr); RG_PROF_M(2); }
#ifdef RG_K1_PROF
    pf_m[3] += ((unsigned long long)nc << 32) | 1u;
#endif
    // new places in the pending list: a pending entry moves up by the candidates ranked before it; a candidate lands
    // behind the pending entries and the candidates ranked before it
    const bool plane = (uint32_t)lane < bm.psize;
    const uint32_t pid = bm.pi & ~kFlagBit;
    uint32_t up = 0, crank = 0, prank = 0;
    for (unsigned long long m = vmask; m; m &= m - 1) {
        const int s = __ffsll((long long)m) - 1;
        const float od = readlane_f(cd, s);
        const uint32_t oi = readlane_u(cid, s);
        const bool before_me = plane && nb_less(od, oi, bm.pd, pid);
        up += before_me ? 1u : 0u;
        crank += nb_less(od, oi, cd, cid) ? 1u : 0u;
        const uint32_t behind = bm.psize - (uint32_t)__popcll(__ballot(before_me));   // pending entries ranked before candidate s
        if (lane == s) prank = behind;
    }
    uint2 *slots = reinterpret_cast<uint2 *>(mscr);
    if (plane) slots[(uint32_t)lane + up] = make_uint2(__float_as_uint(bm.pd), bm.pi);
    if (valid) slots[prank + crank] = make_uint2(__float_as_uint(cd), cid);
    lds_fence();
    bm.psize += nc;
    if ((uint32_t)lane < bm.psize) { const uint2 t = slots[lane]; bm.pd = __uint_as_float(t.x); bm.pi = t.y; }
    lds_fence();
    // eviction (neighbor.h:168-170): the worst entries beyond the capacity fall off, from whichever container holds them
    while (bm.size + bm.psize > bm.cap) {
        bool drop_pending = bm.size == 0;
        if (!drop_pending && bm.psize > 0) {
            const uint2 w = bm.ent[bm.size - 1];
            const float d2 = readlane_f(bm.pd, (int)bm.psize - 1);
            const uint32_t i2 = readlane_u(bm.pi, (int)bm.psize - 1) & ~kFlagBit;
            drop_pending = nb_less(__uint_as_float(w.x), w.y & ~kFlagBit, d2, i2);
        }
        if (drop_pending) --bm.psize;
        else { --bm.size; bm.cur = min(bm.cur, bm.size); }
    }
    RG_PROF_M(1);
}

constexpr uint32_t kCountedBit = 0x80000000u;   // qlog_n[q]: the distinct count of the query's log is already in out_cmps[q]

// Exact number of DISTINCT ids among log[0, n) -- one wave, K4's half-word bucket set (rg_distinct_kernel<true>,
// rg_search.hip) over `tab`: T = 2^tbits words of 8-slot buckets (16-bit remainders of the bijective hash id * odd mod
// 2^id_bits, bucket = its top bits) + T/8 words of exact side table for ids whose bucket is full; logs above 5T/4 ids are
// counted in hash partitions.  fail = the side table filled up (the query is then left to K4).
template <int NL = 4>
__device__ __forceinline__ uint32_t wave_distinct_half(const uint32_t *log, uint32_t n, uint32_t *tab, uint32_t tbits,
                                                       uint32_t id_bits, int lane, bool &fail) {
    const uint32_t T = 1u << tbits, bbits = tbits - 2u, OV = T / 8u;
    uint32_t *side = tab + T;
    const uint32_t cap = (T / 4u) * 5u;
    const uint32_t rbits = id_bits - bbits, hmask = id_bits >= 32u ? 0xffffffffu : (1u << id_bits) - 1u;
    const uint32_t parts = (n + cap - 1u) / cap;
    uint32_t mine = 0;
    bool bad = false;
    for (uint32_t p = 0; p < parts; ++p) {
        for (uint32_t i = (uint32_t)lane * 4u; i < T + OV; i += kWave * 4u) *reinterpret_cast<uint4 *>(tab + i) = make_uint4(~0u, ~0u, ~0u, ~0u);
        lds_fence();
        for (uint32_t i0 = (uint32_t)lane; i0 < n; i0 += kWave * (uint32_t)NL) {
            uint32_t v[NL];   // NL independent loads in flight (the log may be another XCD's: agent-scope loads, past the L1 and the local L2's stale lines), then the inserts
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const uint32_t i = i0 + (uint32_t)u * kWave;
                v[u] = i < n ? __hip_atomic_load(log + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
            }
#pragma nounroll
            for (int u = 0; u < NL; ++u) {   // one copy of the insert code: the values rotate through v[0]
                const uint32_t id = v[0];
#pragma unroll
                for (int w = 0; w + 1 < NL; ++w) v[w] = v[w + 1];
                if (id == 0xffffffffu) continue;
                if (parts > 1u && ((id * 0x85EBCA6Bu) >> 16) % parts != p) continue;
                const uint32_t h = (id * 0x9E3779B1u) & hmask;
                const uint32_t b = h >> rbits, rem = h & ((1u << rbits) - 1u);
                for (;;) {
                    const uint4 t = *reinterpret_cast<const uint4 *>(tab + 4u * b);
                    const uint32_t w4[4] = {t.x, t.y, t.z, t.w};
                    int e = 8;
                    bool found = false;
#pragma unroll
                    for (int k2 = 7; k2 >= 0; --k2) {
                        const uint32_t hv = (k2 & 1) ? w4[k2 >> 1] >> 16 : w4[k2 >> 1] & 0xffffu;
                        found |= hv == rem;
                        if (hv == 0xffffu) e = k2;
                    }
                    if (found) break;
                    if (e == 8) {   // home bucket full: exact side table of full ids
                        uint32_t slot = (id * 0x85EBCA6Bu) >> (32u - (tbits - 3u)), probes = 0;
                        for (;;) {
                            const uint32_t old = atomicCAS(&side[slot], 0xffffffffu, id);
                            if (old == 0xffffffffu) { ++mine; break; }
                            if (old == id) break;
                            slot = (slot + 1u) & (OV - 1u);
                            if (++probes >= OV) { bad = true; break; }
                        }
                        break;
                    }
                    const uint32_t w = e < 2 ? t.x : e < 4 ? t.y : e < 6 ? t.z : t.w;
                    const uint32_t nw = (e & 1) ? (w & 0x0000ffffu) | (rem << 16) : (w & 0xffff0000u) | rem;
                    if (atomicCAS(&tab[4u * b + (uint32_t)(e >> 1)], w, nw) == w) { ++mine; break; }
                }
            }
        }
        lds_fence();
    }
    fail = __any(bad);
    for (int o = 32; o; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o, 64);
    return mine;
}

// DIMC: 0 = any dimension (query staged in LDS), else the compile-time dimension (query in registers)
// BF:   opt-in fast mode, NOT parity: the traversal scores a bf16 copy of the base (4 instead of 7 HBM
//       lines per d = 200 evaluation); at the end the whole beam is re-scored with the exact fp32 routine and the k best
//       by exact (distance, id) are returned, so the reported distances are exact for the returned ids.
//
// Register budget (amdgpu_waves_per_eu): the register-staged gather wants many VGPRs, LDS leaves room for 12 - 16 resident
// queries per CU at L_pq <= 500 -- four waves per SIMD (128 VGPRs) up to four register sets at d = 200, three (170) for the
// two sets of d = 512, two (256) for the eight-set form that wide beams use.
// P.spec == 2 ("multi_expand", opt-in, NOT parity -- SURVEY 8(f-4), speculative multi-expansion): every iteration pops
// the TWO closest unexpanded entries and expands both in one adjacency / visited / gather phase.  The second one is not
// necessarily the node the reference would expand next (a neighbour of the first may have been closer), so the visiting
// order -- and with it, occasionally, the result -- differs; twice the fresh neighbours share one latency chain.
//
// VIS: how the visited set (VisitedList, visited_list_pool.h:8-29) is kept
//   0  exact epoch-tagged words in HBM, test-and-set with one returning atomic per neighbour (a dependent round trip per hop)
//   1  lossy exact-match filter in LDS (+ id log -> K4 for the exact cmps)
//   2  LOOK-AHEAD over the same exact words (round 3; d = 200 / 512, ELL rows without repeated ids): the test is a plain
//      load of the word, the mark a fire-and-forget atomic issued only for fresh neighbours -- so testing has no side
//      effect and can be done EARLY.  Once a hop's candidates are scored, the node the next pop will return is known
//      (the closest of the beam's best unexpanded entry and the new candidates) -- before the inserts and the pop, a
//      quarter of a wide-beam hop.  Its adjacency row is requested there and then, its visited words as soon as the
//      row is in; where a cheap guess made before the gather (the beam's best unexpanded entry: right in about half of
//      the hops) turns out right, the row is already there and the words leave at once.  The next hop then starts
//      with its fresh list a few register operations away: the dependent chain of a hop is the row gather plus whatever
//      of the two small reads the inserts and the pop did not cover, instead of adjacency -> visited -> rows.
//      Exactness: the words of the next node are read after an explicit vmcnt(0) behind the scoring, i.e. after every
//      mark of the current hop has been acknowledged by the L2 (the loads bypass the L1: agent-scope atomic loads).
//      A first version fetched the guessed node's visited words under the gather: with half the guesses wrong the
//      wasted sector reads cost more than the hits gained (profiles/r03/k1_ab_lookahead_10m.jsonl).
//
//   3  EXACT SET IN LDS (round 4; the compute-layout instantiations): eight-entry buckets of 16-bit remainders of a bijective
//      id hash + a side table of full ids -- K4's set, kept by the searching wave itself in the region the filter has.
//      Nothing is forgotten, so nothing is scored twice: cmps is exact as it is counted, without id log, K4 or
//      de-duplicating inserts.  It holds what a narrow beam visits (a few thousand nodes in 10 KiB); a query that outgrows
//      it finishes in the forgetful form and counts its short log itself (SearchParams::vs_side).
//
// GF: gather form of the register-staged instantiations
//   0  rows fetched 16 bytes per lane (global_load_dwordx4), then bounced block by block through a 1-KiB LDS buffer into
//      the layout the scoring routine reads (round 2)
//   1  rows fetched in the COMPUTE layout, one dword per lane and step (round 3): the same bytes per wave instruction
//      reach the texture unit, a row's 128-byte line is covered by two consecutive instructions, and the score is a
//      dozen FMAs on the registers the loads filled -- the LDS round trips of the bounce (a fifth of a wide-beam hop's
//      wave cycles, profiles/r03/k1_phases_10m_*.json) are gone.  Same accumulation order, same bits.
template <bool L2, bool ELL, int R, int VIS, int DIMC, bool BF, int GF = 0>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DIMC == 512 ? (R <= 1 ? 4 : R <= 2 ? 3 : 2) : (R <= 4 ? 4 : 2)))) rg_search_kernel(SearchParams P) {
    static_assert(!BF || (DIMC != 0 && ELL), "fast mode: compile-time dimension, ELL adjacency");
    static_assert(GF == 0 || (DIMC != 0 && !BF), "compute-layout gather: register-staged instantiations");
    static_assert(VIS != 2 || (DIMC != 0 && ELL && !BF), "look-ahead form: register-staged gather over ELL rows");
    static_assert(VIS != 3 || (DIMC != 0 && ELL && !BF), "exact LDS set: register-staged gather over ELL rows");
    constexpr bool EXACT = VIS == 0 || VIS == 2;   // visited words in HBM
    constexpr bool LOOK = VIS == 2;
    constexpr bool LSET = VIS == 3;                // exact visited set in LDS
    constexpr bool LOGS = VIS == 1 || VIS == 3;    // forms that may log the ids they score
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
    // kCand words: candidate c's tail slot (split rows) until its row is requested, then its distance bits
    uint32_t *cand_x = cand_id + kCand;
    uint32_t *mscr = cand_x + kCand;                                      // 128: merge scratch
    Beam bm;
    bm.ent = reinterpret_cast<uint2 *>(mscr + 2 * kWave);                 // L
    bm.cap = P.L;
    // VIS=1: lossy exact-match visited filter (direct mapped, 16-bit remainders of a bijective id hash)
    // id-log staging: ids are appended here and flushed to HBM 64 at a time (256-B aligned full-line stores; small
    // unaligned appends would turn into read-modify-writes at the memory side once the line has left L2)
    uint32_t *logbuf = reinterpret_cast<uint32_t *>(bm.ent + P.L);        // 128
    uint32_t *hb32 = logbuf + (LOGS ? 128 : 0);                           // hub bitmap (P.hub_words words; none when hub_m == 0); forms that log nothing have no log line
    uint16_t *vtab = reinterpret_cast<uint16_t *>(hb32 + P.hub_words);
    const uint32_t vf_up = 32u - P.id_bits;   // hashed id -> top of the word (id_bits >= 1)
    const uint32_t vf_id_mask = P.id_bits >= 32u ? 0xffffffffu : ((1u << P.id_bits) - 1u);
    const uint32_t vf_rem_mask = (1u << P.vf_rem_bits) - 1u;

    uint32_t *vmap = P.visited + (size_t)blockIdx.x * P.vwords;
    // VIS = 3 with P.visited set ("lset_tags"): the nodes the exact LDS set has no room for go to the exact byte tags in HBM
    // instead of being forgotten -- the form stays exact without a log, for beams whose visits outgrow the LDS
    const bool ltags = LSET && P.visited != nullptr;
    uint32_t epoch = (EXACT || ltags) ? P.slot_epoch[blockIdx.x] : 0u;
    unsigned long long tot_n = 0, tot_d = 0;   // in-kernel distinct count: this slot's share of the batch totals
    unsigned long long tot_left = 0;           // VIS = 3: queries of this slot that outgrew the exact set

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
        uint32_t *qlog = (LOGS && P.qlog) ? P.qlog + (size_t)qi * P.logcap : nullptr;
        bool logging = VIS == 1;      // VIS = 3: from the hop on at which the query outgrew the exact set
        bool left = false;            // VIS = 3: some lane found no room for its node (set by visit_set)
        uint32_t logn = 0, lbn = 0;   // ids scored so far / ids waiting in logbuf
        RG_PROF_DECL;
        if constexpr (DIMC != 0) load_query_regs<DIMC>(query, qr, lane);
        else for (uint32_t i = lane; i < P.dim; i += kWave) qv[i] = query[i];
        if constexpr (BF) load_query_regs_bf<DIMC>(query, qb, lane);
        // new visited epoch (VisitedList::reset, visited_list_pool.h:20-26: ++curV, wipe on wrap)
        uint32_t etag = 0;
        if (EXACT || ltags) {
            if (++epoch == (((LOOK && P.vbytes) || ltags) ? 0x100u : 0x10000u)) {
                for (uint32_t w = lane; w < P.vwords; w += kWave) vmap[w] = 0u;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                epoch = 1;
            }
            etag = epoch << 16;
        }
        if (LOOK && P.hub_m) for (uint32_t i = lane; i < P.hub_words; i += kWave) hb32[i] = 0u;     // no hub visited yet
        if (VIS == 1 || LSET || P.vf_front) {   // exact-match filter / exact set: every slot empty; LOOK: the screen's bits clear
            uint32_t *vt32 = reinterpret_cast<uint32_t *>(vtab);
            const bool lsf = LOOK && P.ls_front != 0u;    // LOOK with the exact set in front: set + side empty, then the screen's bits clear
            const uint32_t nset = P.vf_slots / 2u + ((LSET || lsf) ? P.vs_side : 0u);
            for (uint32_t i = lane; i < nset; i += kWave) vt32[i] = (LOOK && !lsf) ? 0u : 0xffffffffu;
            if (lsf) for (uint32_t i = lane; i < P.bl_words; i += kWave) vt32[nset + i] = 0u;
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
        // LDS filter slot / remainder of an id.  x = bijective hash of the id (odd multiplier mod 2^id_bits); the slot is
        // floor(x * slots / 2^id_bits), so the x that share a slot are at most ceil(2^id_bits / slots) <= 2^rem_bits
        // CONSECUTIVE integers, which their low rem_bits bits tell apart: (slot, rem) <-> id is one to one for ANY slot
        // count -- the filter takes whatever LDS the resident queries leave, not the power of two below it (a power of two
        // gives slot = the top bits of x, rem = the rest)
        auto vf_hash = [&](uint32_t id, uint32_t &slot, uint16_t &rem) __attribute__((always_inline)) {
            const uint32_t x = (id * 0x9E3779B1u) & vf_id_mask;
            slot = __umulhi(x << vf_up, P.vf_slots);
            rem = (uint16_t)(x & vf_rem_mask);
        };
        // LOOK form: the same LDS region is a one-hash bit screen in front of the exact words instead -- a bit is set when
        // its node is marked, so a CLEAR bit proves "not visited yet" and the 128-byte line read of the word is not issued
        // at all (a set bit proves nothing: the word decides).  No false "visited", so every output stays exact; what it
        // saves is memory transactions: a wide beam tests 1.8 nodes per node it scores, and more than half of the tests
        // are of nodes never met before.
        // P.ls_front (round 4): the front of the region is an EXACT SET (the layout of VIS = 3: vf_slots entries + vs_side words), the
        // screen the bl_words behind it.  A node the set holds needs no tag at all -- no mark stored, no line read when it is met
        // again (the re-encounters are most of the tag lines a wide beam reads: 22 of 25 per hop at L_pq 500) -- and the tags and
        // the screen keep only the nodes the set had no room for.
        const bool fset = LOOK && P.ls_front != 0u;
        uint32_t *bl32 = reinterpret_cast<uint32_t *>(vtab) + (fset ? P.vf_slots / 2u + P.vs_side : 0u);
        const uint32_t bl_bits = fset ? P.bl_words * 32u : P.vf_slots * 16u;
        const bool screen = LOOK && P.vf_front != 0u;
        auto bl_maybe = [&](uint32_t id) __attribute__((always_inline)) -> bool {
            const uint32_t p = __umulhi(id * 0x9E3779B1u, bl_bits);
            return (bl32[p >> 5] >> (p & 31u)) & 1u;
        };
        auto bl_set = [&](uint32_t id) __attribute__((always_inline)) {
            const uint32_t p = __umulhi(id * 0x9E3779B1u, bl_bits);
            (void)__hip_atomic_fetch_or(&bl32[p >> 5], 1u << (p & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // visited test-and-set of this lane's neighbour (:2378, :2385); same-hop duplicates are resolved by the atomic's
        // order (VIS = 0) or by the merge's de-duplication (VIS = 1)
        // neighbour word of an adjacency row -> id, and whether the LDS filter keeps an entry for it (see id_mask)
        const uint32_t idm = P.id_mask;
        auto keeps = [&](uint32_t idw) __attribute__((always_inline)) -> bool {
            return idm == 0xffffffffu || ((idw >> 24) & 15u) >= P.vf_min_indeg;     // (low nibble of the top byte: min(15, in-degree))
        };
        // hub bits (SearchParams::hub_m): is the neighbour word's node a hub of this launch's bitmap / test-and-set of its bit
        // (a returning LDS atomic: two lanes of one hop that bring the same hub -- rows that name a node twice -- get one "fresh")
        auto hub_of = [&](uint32_t idw) __attribute__((always_inline)) -> bool {
            return LOOK && P.hub_m != 0u && 8u + (idw >> 28) <= P.hub_m;
        };
        auto hub_visit = [&](uint32_t id) __attribute__((always_inline)) -> bool {
            const uint32_t p = (id * 0x9E3779B1u) >> (32u - P.hub_m);
            const uint32_t bit = 1u << (p & 31u);
            return !(__hip_atomic_fetch_or(&hb32[p >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit);
        };
        // exact set in LDS (VIS = 3; VIS = 2 with P.ls_front: in FRONT of the byte tags).  bucket = floor(x * buckets / 2^id_bits) of the
        // bijective hash x, entry = the low bits of x that tell the x of one bucket apart (vf_hash with slots = buckets):
        // (bucket, entry) <-> id.  Found -> visited.  Not found -> the first empty entry takes it (CAS on its word: two lanes of
        // one hop that bring the same node meet here, and exactly one of them is fresh); a full bucket sends the node to the side
        // table: buckets of four full ids (one ds_read_b128 each), at most four of them from the id's hash on.  A node is only
        // ever placed within that reach, so a lookup that finds it nowhere there has seen everything; no room within it = the
        // set cannot take the node.  (The first version probed single words linearly: a full table cost a CAS round trip per word
        // and lookup -- the cliff of profiles/r04/k1_ab_box14_lset_plan.txt.)
        // ls_visit: 0 = visited, 1 = inserted now (fresh), 2 = not in the set and no room for it.  ls_lookup: found or not.
        auto ls_visit = [&](uint32_t id, bool insert) __attribute__((always_inline)) -> int {
            const uint32_t x = (id * 0x9E3779B1u) & vf_id_mask;
            const uint32_t b = __umulhi(x << vf_up, P.vf_slots >> 3);
            const uint16_t rem = (uint16_t)(x & vf_rem_mask);
            uint32_t *bk = reinterpret_cast<uint32_t *>(vtab) + 4u * b;
            for (;;) {
                const uint4 t = *reinterpret_cast<const uint4 *>(bk);
                const uint32_t w4[4] = {t.x, t.y, t.z, t.w};
                int e = 8;
                bool found = false;
#pragma unroll
                for (int k2 = 7; k2 >= 0; --k2) {
                    const uint32_t hv = (k2 & 1) ? w4[k2 >> 1] >> 16 : w4[k2 >> 1] & 0xffffu;
                    found |= hv == (uint32_t)rem;
                    if (hv == 0xffffu) e = k2;
                }
                if (found) return 0;
                if (e == 8) {
                    uint32_t *side = reinterpret_cast<uint32_t *>(vtab) + (P.vf_slots >> 1);
                    const uint32_t nsb = P.vs_side >> 2;
                    uint32_t sb = __umulhi(id * 0x85EBCA6Bu, nsb);
                    for (int pr = 0; pr < 4; ++pr) {
                        uint32_t *sp = side + 4u * sb;
                        for (;;) {
                            const uint4 sv = *reinterpret_cast<const uint4 *>(sp);
                            if (sv.x == id || sv.y == id || sv.z == id || sv.w == id) return 0;                      // visited
                            const int se = sv.x == 0xffffffffu ? 0 : sv.y == 0xffffffffu ? 1 : sv.z == 0xffffffffu ? 2 : sv.w == 0xffffffffu ? 3 : 4;
                            if (se == 4) break;                                                              // full: the next bucket
                            if (!insert) return 2;             // (a free entry in reach: the node is not in the set)
                            const uint32_t old = atomicCAS(sp + se, 0xffffffffu, id);
                            if (old == 0xffffffffu) return 1;
                            if (old == id) return 0;                                                         // another lane of this hop brought it
                        }
                        if (++sb == nsb) sb = 0;
                    }
                    return 2;
                }
                if (!insert) return 2;
                const uint32_t w = e < 2 ? t.x : e < 4 ? t.y : e < 6 ? t.z : t.w;
                const uint32_t nw = (e & 1) ? (w & 0x0000ffffu) | ((uint32_t)rem << 16) : (w & 0xffff0000u) | (uint32_t)rem;
                if (atomicCAS(&bk[e >> 1], w, nw) == w) return 1;
            }
        };
        auto visit_set = [&](uint32_t id, bool have, bool keep, bool hub) __attribute__((always_inline)) -> bool {
            bool fresh = false;
            if (LOOK && hub) {
                if (have) fresh = hub_visit(id);
                if (LOOK && P.vbytes) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (as the tag path below leaves it)
            } else if (LSET) {
                if (have) {
                    const int r = ls_visit(id, true);
                    if (r == 1) fresh = true;
                    else if (r == 2 && ltags) {                    // no room: the node's epoch byte in HBM decides and remembers
                        uint8_t *t = reinterpret_cast<uint8_t *>(vmap) + id;
                        fresh = __hip_atomic_load(t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint8_t)epoch;
                        if (fresh) __hip_atomic_store(t, (uint8_t)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if (r == 2) { fresh = true; left = true; }      // no room: scored, not remembered
                }
            } else if (VIS == 1) {
                // exact-match lookup: a hit proves "visited"; a miss is treated as fresh (may re-score a node whose
                // entry was overwritten -- harmless for the beam, see beam_insert<true>)
                if (have) {
                    fresh = true;
                    if (keep) {
                        uint32_t slot; uint16_t rem;
                        vf_hash(id, slot, rem);
                        fresh = vtab[slot] != rem;
                        if (fresh) vtab[slot] = rem;
                    }
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
                if (!LOOK && P.vf_front && keep) {
                    uint32_t slot; uint16_t rem;
                    vf_hash(id, slot, rem);
                    known = vtab[slot] == rem;
                    if (!known) vtab[slot] = rem;
                }
                if (LOOK && P.vbytes) {                    // byte tags (general path of the LOOK form: long rows, shared first hop)
                    const int r = fset ? ls_visit(id, true) : 2;
                    if (r == 1) fresh = true;
                    else if (r == 2) {
                        uint8_t *t = reinterpret_cast<uint8_t *>(vmap) + id;
                        fresh = __hip_atomic_load(t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint8_t)epoch;
                        if (fresh) __hip_atomic_store(t, (uint8_t)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next hop's tests leave without a wait of their own
                    if (screen && fresh && r == 2) bl_set(id);
                    return fresh;
                } else if (!known) {
                    uint32_t *w = &vmap[id >> 4];
                    const uint32_t bit = 1u << (id & 15u);
                    atomicMax(w, etag);                    // stale epoch -> word becomes (epoch, no bits)
                    const uint32_t old = atomicOr(w, bit); // same address, same lane: ordered behind the max
                    fresh = !(old & bit);
                }
                if (screen && fresh) bl_set(id);
            }
            return fresh;
        };
        // id log (VIS = 1): n ids from cand_id[off ..) join the LDS line buffer; a full 64-id line leaves as one aligned
        // 256-B store.  The store is issued where no gather is outstanding: in front of gathers it would sit at the head
        // of the vmcnt queue and put its completion latency on the critical path of the first counted wait.
        auto log_put = [&](uint32_t off, uint32_t n) __attribute__((always_inline)) {
            if (LOGS && qlog) {
                if ((uint32_t)lane < n) logbuf[lbn + lane] = cand_id[off + lane];
                lbn += n;
            }
        };
        auto log_append = [&](uint32_t off, uint32_t n) __attribute__((always_inline)) {
            log_put(off, n);
            logn += n;
        };
        auto log_flush = [&]() __attribute__((always_inline)) {
            if (LOGS && qlog && lbn >= (uint32_t)kWave) {
                lds_fence();
                const uint32_t pos = logn - lbn;                       // ids already flushed (multiple of 64)
                const uint32_t v = logbuf[lane], rest = logbuf[kWave + lane];
                lds_fence();
                lbn -= kWave;
                if ((uint32_t)lane < lbn) logbuf[lane] = rest;
                // agent-scope store: the counter of this log may run on another XCD, whose L2 is not this one's
                if (pos + lane < P.logcap) __hip_atomic_store(&qlog[pos + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lds_fence();
            }
        };
        // gather + score (:2387) of cand_id[0 .. n) into cand_d, 4 rows per pass.
        //  * compile-time dimension (the BASELINE shapes): REGISTER-STAGED.  A batch of R passes (4R rows) is fetched with
        //    plain global_load_dwordx4 into R register sets -- all of them in flight together, one HBM latency per batch --
        //    and each set is then bounced, 64-element block by block, through ONE 1-KiB LDS buffer in the slot layout the scoring routine reads (the layout
        //    the LDS-DMA path produces: lane l's 16 bytes at 16 l inside each 1-KiB block).  Rows in flight cost VGPRs, of
        //    which a latency-bound wave has plenty, instead of LDS, which is what limits resident queries at large L_pq.
        //  * otherwise (any dimension, or the bf16 fast mode): LDS-DMA into a ring of R staging buffers; pass p is consumed
        //    once only the loads of the passes issued after it are still outstanding.
        // `hook` runs once, right behind the load instructions of the first batch (the look-ahead form issues its early
        // visited-word loads there: behind the rows in the memory queue, ahead of the first wait)
        auto gather_list = [&](uint32_t n, auto hook, auto hooked) __attribute__((always_inline)) {
            const uint32_t npass = (n + 3u) >> 2;
            if constexpr (DIMC != 0 && !BF) {
                typedef v4f_t v4f;
                constexpr int NFULL = DIMC / 64, REM = DIMC & 63;
                const int jsrc = ((lane & 15) - 4 * g) & 15;
                const uint32_t rid0 = cand_id[0];
                const bool split = REM != 0 && P.tail_off != nullptr;
                const uint32_t tix0 = split ? cand_x[0] : rid0;
                // STREAM (round 3, knob "gather_roll", default on): a hop with more than 4R fresh rows used to pay one memory
                // latency per batch of R sets.  In the streamed form set j is re-loaded with the rows of pass p + R as soon
                // as pass p has been scored out of it, so 4R rows stay in flight from the first pass of a hop to its last
                // instead of draining at every batch boundary.  The loads of the steady-state loop are unconditional (passes
                // beyond the list re-read row 0, as idle sets do), so the waits the compiler puts in front of every set stay
                // exact counts; the last group is scored by an epilogue that loads nothing.
                auto batch = [&](uint32_t p0, auto first_batch, auto stream) __attribute__((always_inline)) {
                    constexpr bool STREAM = decltype(stream)::value;
                    constexpr int NV = GF == 1 ? DIMC / 16 : 1;
                    v4f rv[GF == 1 ? 1 : R][NFULL];
                    float rw[GF == 1 ? R : 1][NV];
                    float t8[R];
                    uint32_t rid[R], tix[R];
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        const uint32_t c = 4 * (p0 + j) + g;
                        // idle groups and idle sets re-read a row that is being fetched anyway: every set is loaded
                        // unconditionally so that the waits in front of the sets are exact counts (a set behind a branch
                        // makes the compiler wait for ALL sets before the first one is scored: -10 % measured)
                        rid[j] = c < n ? cand_id[c] : rid0;
                        if constexpr (REM != 0) tix[j] = split ? (c < n ? cand_x[c] : tix0) : rid[j];
                    }
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        if constexpr (GF == 1) {
                            const float *src = P.base + (size_t)rid[j] * P.stride + (lane & 15);
#pragma unroll
                            for (int t = 0; t < NV; ++t) rw[j][t] = src[16 * t];
                        } else {
                            const float *src = P.base + (size_t)rid[j] * P.stride + 4 * jsrc;
#pragma unroll
                            for (int b = 0; b < NFULL; ++b) rv[j][b] = *reinterpret_cast<const v4f *>(src + 64 * b);
                        }
                        // the 8-wide tail: lane a's own element 192 + (a & 7), straight into the register that scores it
                        if constexpr (REM != 0) t8[j] = P.tail_base[(size_t)tix[j] * P.tail_stride + (lane & 7)];
                        else t8[j] = 0.0f;
                    }
                    if constexpr (decltype(first_batch)::value) hook();
                    if constexpr (STREAM) {
                        for (; p0 + R < npass; p0 += R) {   // every pass of this group exists and has a successor set to load
#pragma unroll
                            for (int j = 0; j < R; ++j) {
                                RG_PROF(6);
                                float d;
                                if constexpr (GF == 1) d = regs_score_q<L2, DIMC>(rw[j], t8[j], qr);
                                else d = bounce_score_q<L2, DIMC>(stage, rv[j], t8[j], qr, lane);
                                const uint32_t c = 4 * (p0 + j) + g;
                                if (c < n && (lane & 15) == 0) cand_x[c] = __float_as_uint(d);
                                lds_fence();
                                RG_PROF(3);
                                // pass p0 + R + j takes the registers over (scores land in cand_x[.. 4 (p0 + R)): the tail
                                // slots of the passes still to be loaded are intact)
                                const uint32_t c2 = 4 * (p0 + R + j) + g;
                                const uint32_t rid2 = c2 < n ? cand_id[c2] : rid0;
                                uint32_t tix2 = rid2;
                                if constexpr (REM != 0) tix2 = split ? (c2 < n ? cand_x[c2] : tix0) : rid2;
                                if constexpr (GF == 1) {
                                    const float *src = P.base + (size_t)rid2 * P.stride + (lane & 15);
#pragma unroll
                                    for (int t = 0; t < NV; ++t) rw[j][t] = src[16 * t];
                                } else {
                                    const float *src = P.base + (size_t)rid2 * P.stride + 4 * jsrc;
#pragma unroll
                                    for (int b = 0; b < NFULL; ++b) rv[j][b] = *reinterpret_cast<const v4f *>(src + 64 * b);
                                }
                                if constexpr (REM != 0) t8[j] = P.tail_base[(size_t)tix2 * P.tail_stride + (lane & 7)];
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        if (p0 + j < npass) {
                            RG_PROF(6);
                            float d;
                            if constexpr (GF == 1) d = regs_score_q<L2, DIMC>(rw[j], t8[j], qr);
                            else d = bounce_score_q<L2, DIMC>(stage, rv[j], t8[j], qr, lane);
                            const uint32_t c = 4 * (p0 + j) + g;
                            if (c < n && (lane & 15) == 0) cand_x[c] = __float_as_uint(d);
                            lds_fence();
                            RG_PROF(3);
                        }
                    }
                };
                if (P.roll) {
                    batch(0u, hooked, std::true_type{});
                } else if constexpr (decltype(hooked)::value) {
                    batch(0u, std::true_type{}, std::false_type{});
                    for (uint32_t p0 = R; p0 < npass; p0 += R) batch(p0, std::false_type{}, std::false_type{});
                } else {
                    for (uint32_t p0 = 0; p0 < npass; p0 += R) batch(p0, std::false_type{}, std::false_type{});
                }
            } else {
                const uint32_t lpp = BF ? (uint32_t)NB : loads_per_pass(P.dim);
                for (uint32_t p = 0; p < (uint32_t)R && p < npass; ++p) {
                    const uint32_t c = 4 * p + g;
                    const uint32_t rid = c < n ? cand_id[c] : 0u;
                    issue(rid, c < n, stage + (size_t)p * P.stage_floats);
                }
                if constexpr (decltype(hooked)::value) hook();
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
                    if (c < n && (lane & 15) == 0) cand_x[c] = __float_as_uint(d);
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

        auto no_hook = []() __attribute__((always_inline)) {};
        // entry point: scored and queued, not marked visited (index_bipartite.cpp:2338-2352)
        float epd;
        const float *front = (P.front_scores && !BF) ? P.front_scores + (size_t)qi * P.front_stride : nullptr;
        if (front) epd = front[0];
        else if constexpr (DIMC != 0 && !BF) {
            if (lane == 0) { cand_id[0] = P.ep; cand_x[0] = P.ep_tail; }
            lds_fence();
            gather_list(1, no_hook, std::false_type{});
            epd = __uint_as_float(cand_x[0]);
            lds_fence();
        } else {
            issue(P.ep, g == 0, stage);
            gather_wait(0);
            epd = score(stage);
        }
        if (lane == 0) bm.ent[0] = make_uint2(__float_as_uint(epd), P.ep);
        bm.size = 1;
        bm.cur = 0;
        bm.psize = 0;
        bm.pd = 0.0f;
        bm.pi = 0u;
        wave_sync();

        uint32_t cmps = 0, hops = 0;
        // one expansion (:2368-2399): neighbours of `node` 64 at a time -- visited test-and-set, gather + score, insert
        auto expand = [&](uint32_t node, uint32_t first, uint32_t toff) __attribute__((always_inline)) {
            uint32_t deg;
            const uint32_t *list;
            if (ELL) {
                deg = readlane_u(first, 0);
                list = P.ell + (size_t)node * P.ell_stride + 1;
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
                const bool keep = keeps(id);
                const bool hub = ELL && hub_of(id);
                if (ELL) id &= idm;
                if (build && id == tgt) have = false;
                const bool fresh = visit_set(id, have, keep, hub);
                const unsigned long long fm = __ballot(fresh);
                const uint32_t n = __popcll(fm);
                RG_PROF_CNT(0, 1); RG_PROF_CNT(1, n);
                if (n == 0) { RG_PROF(2); continue; }
                const bool from_front = front != nullptr && hops == 1u && node == P.ep;   // the batch-wide scores of ep's neighbours
                if (fresh) {
                    const uint32_t pos = __popcll(fm & ((1ull << lane) - 1ull));
                    cand_id[pos] = id;
                    if (from_front) cand_x[pos] = __float_as_uint(front[1u + c0 + lane]);
                    else if (P.tail_off) cand_x[pos] = toff + c0 + lane;   // the neighbour's place in the adjacency order
                }
                lds_fence();
                if constexpr (LSET) {
                    // a lane found no room for its node: from this hop on the set is incomplete -- the ids scored are logged (this
                    // hop's too: none of them was scored before), inserts de-duplicate, and the log's distinct ids join cmps at the end
                    if (!logging && __any(left)) logging = true;
                    // (both counters advance by a selected VALUE: written as "if (logging) logn += n; else cmps += n;" the compiler
                    // sinks the two additions into one through a selected ADDRESS, which puts both counters in scratch memory --
                    // a load, a vmcnt(0) and a store per hop)
                    const uint32_t n_log = logging ? n : 0u;
                    if (logging) log_put(0, n);
                    logn += n_log;
                    cmps += n - n_log;                                     // :2397
                } else {
                    log_append(0, n);
                    cmps += n;                                             // :2397
                }
                RG_PROF(2);
                if (from_front) log_flush();
                else
                // the id-log line of this hop leaves BEHIND the row loads (knob "log_early", default): a store issued after the
                // gather has been consumed sits in front of the next hop's adjacency load in the in-order memory counter;
                // behind the row loads it is covered by the gather's own wait.  (Measured: within 0.4 % either way,
                // profiles/r03/k1_ab_box14.jsonl -- the write acknowledgement was never on the critical path.)
                if (VIS == 1 && P.log_early) gather_list(n, [&]() __attribute__((always_inline)) { log_flush(); }, std::true_type{});
                else {
                    gather_list(n, no_hook, std::false_type{});
                    log_flush();
                }
                // queue inserts (:2398)
                const float cd = (uint32_t)lane < n ? __uint_as_float(cand_x[lane]) : 0.0f;
                const uint32_t cid = (uint32_t)lane < n ? cand_id[lane] : 0u;
                lds_fence();
                RG_PROF(3);
                if constexpr (LSET) {
                    if (logging) beam_insert<true>(bm, cd, cid, (uint32_t)lane < n, P.ep, lane, mscr RG_PROF_MERGE);
                    else beam_insert<false>(bm, cd, cid, (uint32_t)lane < n, P.ep, lane, mscr RG_PROF_MERGE);
                } else beam_insert<VIS == 1>(bm, cd, cid, (uint32_t)lane < n, P.ep, lane, mscr RG_PROF_MERGE);
                RG_PROF(4);
            }
        };
        RG_PROF(5);
        if constexpr (LOOK) {
            // ---- look-ahead form over the exact visited words (see the template's comment).  Carried from hop to hop:
            // the adjacency row (two reads: up to 126 neighbours) and the visited words of the node the NEXT pop will return.
            constexpr uint32_t NONE = 0xffffffffu;
            uint32_t la_node = NONE, la_a = 0, la_b = 0, la_toff = 0, la_wa = 0, la_wb = 0;
            const bool two = P.ell_stride > 64u;                           // rows may need a second read
            auto row_a = [&](uint32_t nd_) __attribute__((always_inline)) { return (uint32_t)lane < P.ell_stride ? P.ell[(size_t)nd_ * P.ell_stride + lane] : 0u; };
            auto row_b = [&](uint32_t nd_) __attribute__((always_inline)) { return two && 64u + (uint32_t)lane < P.ell_stride ? P.ell[(size_t)nd_ * P.ell_stride + 64u + lane] : 0u; };
            // visited words of a row's neighbours: ids 0..62 sit in words 1..63 of the first read, ids 63.. in the second
            auto words_of = [&](uint32_t fa, uint32_t fb, uint32_t &wa, uint32_t &wb) __attribute__((always_inline)) {
                const uint32_t dg = readlane_u(fa, 0);
                const uint32_t wsh = (uint32_t)__shfl_down((int)fa, 1, 64);   // (every lane takes part: never inside a short-circuit condition)
                const uint32_t ia = wsh & idm;
                if (screen || P.vbytes) {
                    // only the lanes whose screen bit is set read their word; 0 = "stale epoch" = fresh for the others
                    lds_fence();
                    // (front set: a node it holds needs no tag; one it does not hold is tested whether or not the set will take it)
                    // (a hub has no tag: its bit in the LDS bitmap is its visited flag)
                    bool ma = (uint32_t)lane < min(dg, 63u) && !hub_of(wsh) && (!screen || bl_maybe(ia));
                    bool mb = dg > 63u && 63u + (uint32_t)lane < dg && !hub_of(fb) && (!screen || bl_maybe(fb & idm));
                    if (fset) {
                        if (ma) ma = ls_visit(ia, false) != 0;
                        if (mb) mb = ls_visit(fb & idm, false) != 0;
                    }
                    wa = 0u; wb = 0u;
                    if (P.vbytes) {
                        const uint8_t *t = reinterpret_cast<const uint8_t *>(vmap);
                        if (ma) wa = __hip_atomic_load(t + ia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (mb) wb = __hip_atomic_load(t + (fb & idm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        if (ma) wa = __hip_atomic_load(&vmap[ia >> 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (mb) wb = __hip_atomic_load(&vmap[(fb & idm) >> 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return;
                }
                wa = __hip_atomic_load(&vmap[((uint32_t)lane < min(dg, 63u) ? ia : 0u) >> 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wb = 0u;
                if (dg > 63u) wb = __hip_atomic_load(&vmap[(63u + (uint32_t)lane < dg ? (fb & idm) : 0u) >> 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            while (beam_has_unexpanded(bm, lane)) {                        // has_unexpanded_node, :2356
                const uint2 popped = beam_pop(bm, lane);                   // :2358
                const uint32_t node = popped.y;
                ++hops;                                                    // :2366
                const bool hit = node == la_node;
                uint32_t fa, fb, toff, wa, wb;
                if (hit) { fa = la_a; fb = la_b; toff = la_toff; wa = la_wa; wb = la_wb; }
                else {
                    fa = row_a(node); fb = row_b(node);
                    toff = P.tail_off ? P.tail_off[node] : 0u;
                }
                RG_PROF(0);
                const uint32_t deg = readlane_u(fa, 0);
                if (deg > 126u || (front != nullptr && hops == 1u)) {   // longer than two reads (or the shared first hop): the general path
                    expand(node, fa, toff);
                    la_node = NONE;
                    continue;
                }
                if (!hit) words_of(fa, fb, wa, wb);
#ifdef RG_K1_PROF
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                RG_PROF(1);
                RG_PROF_CNT(5, deg); RG_PROF_CNT(6, hit ? 1 : 0);
                const uint32_t wA = (uint32_t)__shfl_down((int)fa, 1, 64), idA = wA & idm, idB = fb & idm;
                const bool haveA = (uint32_t)lane < min(deg, 63u), haveB = 63u + (uint32_t)lane < deg;
                const uint32_t bitA = 1u << (idA & 15u), bitB = 1u << (idB & 15u);
                // :2378 (a word / byte the screen spared reads 0, which no epoch equals)
                const bool vb = P.vbytes != 0u;
                // front set first: 0 = it holds the node (visited), 1 = it took the node now (fresh, no tag), 2 = the tags decide
                // hubs first: the bit of the LDS bitmap decides and remembers (3 = hub; no set entry, no tag, no screen bit)
                int rA = 2, rB = 2;
                bool fhA = false, fhB = false;
                if (haveA && hub_of(wA)) { rA = 3; fhA = hub_visit(idA); }
                if (haveB && hub_of(fb)) { rB = 3; fhB = hub_visit(idB); }
                if (fset) {
                    if (haveA && rA != 3) rA = ls_visit(idA, true);
                    if (haveB && rB != 3) rB = ls_visit(idB, true);
                }
                const bool freshA = haveA && (rA == 3 ? fhA : (rA == 1 || (rA == 2 && (vb ? wa != epoch : !((wa >> 16) == epoch && (wa & bitA))))));
                const bool freshB = haveB && (rB == 3 ? fhB : (rB == 1 || (rB == 2 && (vb ? wb != epoch : !((wb >> 16) == epoch && (wb & bitB))))));
                if (vb) {                        // :2385 as a plain byte store: no line is fetched for it, none comes back
                    uint8_t *t = reinterpret_cast<uint8_t *>(vmap);
                    if (freshA && rA == 2) { __hip_atomic_store(t + idA, (uint8_t)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (screen) bl_set(idA); }
                    if (freshB && rB == 2) { __hip_atomic_store(t + idB, (uint8_t)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (screen) bl_set(idB); }
                } else {
                if (freshA && rA != 3) {         // :2385, fire and forget
                    uint32_t *w = &vmap[idA >> 4];
                    (void)__hip_atomic_fetch_max(w, etag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // stale epoch -> (epoch, no bits)
                    (void)__hip_atomic_fetch_or(w, bitA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // same address: behind the max
                    if (screen) bl_set(idA);
                }
                if (freshB && rB != 3) {
                    uint32_t *w = &vmap[idB >> 4];
                    (void)__hip_atomic_fetch_max(w, etag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    (void)__hip_atomic_fetch_or(w, bitB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (screen) bl_set(idB);
                }
                }
                const unsigned long long fmA = __ballot(freshA), fmB = __ballot(freshB);
                const uint32_t nA = __popcll(fmA), n = nA + (uint32_t)__popcll(fmB);
                RG_PROF_CNT(0, 1); RG_PROF_CNT(1, n);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (freshA) { const uint32_t pos = __popcll(fmA & below); cand_id[pos] = idA; if (P.tail_off) cand_x[pos] = toff + lane; }
                if (freshB) { const uint32_t pos = nA + __popcll(fmB & below); cand_id[pos] = idB; if (P.tail_off) cand_x[pos] = toff + 63u + lane; }
                lds_fence();
                cmps += n;                                                 // :2397
                // a cheap early guess: the entry the next pop returns unless one of this hop's candidates is closer (right
                // in about half of the hops); only its adjacency row is fetched on the guess -- one small read under the gather
                float ed = 0.0f;
                uint32_t en = node, ea = 0, eb = 0, etoff = 0;
                const bool ev = beam_peek(bm, lane, ed, en);
                const bool guess = ev && P.look == 1u;
                if (guess) { ea = row_a(en); eb = row_b(en); etoff = P.tail_off ? P.tail_off[en] : 0u; }
                RG_PROF(2);
                if (n) gather_list(n, no_hook, std::false_type{});
                // the candidates, one per lane and chunk of 64
                const bool cvA = (uint32_t)lane < n, cvB = 64u + (uint32_t)lane < n;
                const float cdA = cvA ? __uint_as_float(cand_x[lane]) : 0.0f, cdB = cvB ? __uint_as_float(cand_x[64 + lane]) : 0.0f;
                const uint32_t ciA = cvA ? cand_id[lane] : 0u, ciB = cvB ? cand_id[64 + lane] : 0u;
                lds_fence();
                RG_PROF(3);
                // the node the next pop WILL return: the closest of the guess and this hop's candidates (all of them are
                // unexpanded; the entry point, whose second insert is dropped, is left out).  Known here, before the inserts
                // and the pop, its adjacency row and then its visited words travel under them.
                uint32_t bo = 0xffffffffu, bi = 0xffffffffu;             // best candidate: (ordered distance bits, id)
                {
                    const uint32_t ua = __float_as_uint(cdA), ub = __float_as_uint(cdB);
                    const uint32_t oa = cvA && ciA != P.ep ? ((ua & 0x80000000u) ? ~ua : (ua | 0x80000000u)) : 0xffffffffu;
                    const uint32_t ob = cvB && ciB != P.ep ? ((ub & 0x80000000u) ? ~ub : (ub | 0x80000000u)) : 0xffffffffu;
                    bo = wave_min_u32(min(oa, ob));
                    if (bo != 0xffffffffu) bi = wave_min_u32(min(oa == bo ? ciA : 0xffffffffu, ob == bo ? ciB : 0xffffffffu));
                }
                uint32_t xn = ev ? en : NONE;
                if (bi != 0xffffffffu) {
                    const uint32_t ue = __float_as_uint(ed), oe = (ue & 0x80000000u) ? ~ue : (ue | 0x80000000u);
                    if (!ev || bo < oe || (bo == oe && bi < en)) xn = bi;
                }
                // every mark of this hop has been acknowledged before the words of the next node are read
                // (round 5: the same launch without this wait returns the same bits and runs level -- 80.4 / 78.3 / 70.3 / 60.2 against 80.3 /
                // 78.3 / 69.8 / 60.2 % of 8 TB/s at L_pq 300 ... 2000, profiles/r05/k1_ab_box4_no_ack_wait.txt: the wait is covered by the
                // early guess's row load, which the next step needs anyway; it stays)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bool words_out = false;
                if (xn == NONE) la_node = NONE;
                else if (guess && xn == en) { la_node = en; la_a = ea; la_b = eb; la_toff = etoff; words_of(la_a, la_b, la_wa, la_wb); words_out = true; }
                else { la_node = xn; la_a = row_a(xn); la_b = row_b(xn); la_toff = P.tail_off ? P.tail_off[xn] : 0u; }
                if (n) beam_insert<false>(bm, cdA, ciA, cvA, P.ep, lane, mscr RG_PROF_MERGE);
                if (n > 64u) beam_insert<false>(bm, cdB, ciB, cvB, P.ep, lane, mscr RG_PROF_MERGE);
                if (la_node != NONE && !words_out) words_of(la_a, la_b, la_wa, la_wb);   // its row left before the inserts
                RG_PROF(4);
            }
        } else
        while (beam_has_unexpanded(bm, lane)) {                            // has_unexpanded_node, :2356
            const uint2 popped = beam_pop(bm, lane);                       // :2358
            const uint32_t node = popped.y;
            if (build && lane == 0 && hops < P.exp_cap) P.out_exp[(size_t)qi * P.exp_cap + hops] = popped;   // full_retset, :1319
            ++hops;                                                        // :2366
            uint32_t first = 0;
            if (ELL) first = (uint32_t)lane < P.ell_stride ? P.ell[(size_t)node * P.ell_stride + lane] : 0u;
            uint32_t toff = 0;
            if (ELL && P.tail_off) toff = P.tail_off[node];
            if (ELL && P.spec == 2u && beam_has_unexpanded(bm, lane)) {
                // ---- multi_expand (opt-in, NOT parity): the runner-up is expanded in the same phase
                const uint2 popped2 = beam_pop(bm, lane);
                const uint32_t node2 = popped2.y;
                ++hops;
                const uint32_t first2 = (uint32_t)lane < P.ell_stride ? P.ell[(size_t)node2 * P.ell_stride + lane] : 0u;
                const uint32_t toff2 = P.tail_off ? P.tail_off[node2] : 0u;
                RG_PROF(0);
                const uint32_t deg = readlane_u(first, 0), deg2 = readlane_u(first2, 0);
                if (deg <= 63u && deg2 <= 63u) {
                    const uint32_t wA = (uint32_t)__shfl_down((int)first, 1, 64), wB = (uint32_t)__shfl_down((int)first2, 1, 64);
                    const uint32_t idA = wA & idm, idB = wB & idm;
#ifdef RG_K1_PROF
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                    RG_PROF(1);
                    const bool freshA = visit_set(idA, (uint32_t)lane < deg, keeps(wA), hub_of(wA));
                    const bool freshB = visit_set(idB, (uint32_t)lane < deg2, keeps(wB), hub_of(wB));     // after A's marks: a common neighbour counts once
                    const unsigned long long fmA = __ballot(freshA), fmB = __ballot(freshB);
                    const uint32_t nA = __popcll(fmA), nB = __popcll(fmB);
                    const unsigned long long below = (1ull << lane) - 1ull;
                    if (freshA) { const uint32_t pos = __popcll(fmA & below); cand_id[pos] = idA; if (P.tail_off) cand_x[pos] = toff + lane; }
                    if (freshB) { const uint32_t pos = nA + __popcll(fmB & below); cand_id[pos] = idB; if (P.tail_off) cand_x[pos] = toff2 + lane; }
                    lds_fence();
                    log_append(0, nA);
                    log_flush();
                    log_append(nA, nB);
                    cmps += nA + nB;
                    RG_PROF_CNT(0, 2); RG_PROF_CNT(1, nA + nB);
                    RG_PROF(2);
                    if (nA + nB) gather_list(nA + nB, no_hook, std::false_type{});
                    log_flush();
                    const float cdA = (uint32_t)lane < nA ? __uint_as_float(cand_x[lane]) : 0.0f, cdB = (uint32_t)lane < nB ? __uint_as_float(cand_x[nA + lane]) : 0.0f;
                    const uint32_t ciA = (uint32_t)lane < nA ? cand_id[lane] : 0u, ciB = (uint32_t)lane < nB ? cand_id[nA + lane] : 0u;
                    lds_fence();
                    RG_PROF(3);
                    if (nA) beam_insert<true>(bm, cdA, ciA, (uint32_t)lane < nA, P.ep, lane, mscr RG_PROF_MERGE);
                    if (nB) beam_insert<true>(bm, cdB, ciB, (uint32_t)lane < nB, P.ep, lane, mscr RG_PROF_MERGE);
                    RG_PROF(4);
                } else {
                    expand(node, first, toff);
                    expand(node2, first2, toff2);
                }
                continue;
            }
            RG_PROF(0);
            expand(node, first, toff);
        }

        // results (:2408-2418): the first k entries of the merged beam
        beam_flush(bm, lane, mscr);
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
        if (LOGS && qlog && lbn) {   // tail of the id log
            lds_sync();
            const uint32_t pos = logn - lbn;
            if ((uint32_t)lane < lbn && pos + lane < P.logcap) __hip_atomic_store(&qlog[pos + lane], logbuf[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef RG_K1_PROF
        RG_PROF(5);
        if (P.prof && lane == 0) {
            for (int i = 0; i < 8; ++i) { P.prof[(size_t)qi * 24 + i] = pf_acc[i]; P.prof[(size_t)qi * 24 + 8 + i] = pf_cnt[i]; }
            for (int i = 0; i < 4; ++i) P.prof[(size_t)qi * 24 + 16 + i] = pf_m[i];
        }
#endif
        uint32_t logn_out = logn;
        if (LSET && !cmps_only && !build) {
            const uint32_t pre = cmps;
            if (logging) {    // the query outgrew the exact set: cmps so far + the distinct ids of what was scored since
                bool bad = true;
                uint32_t distinct = 0;
                if (qlog && P.count_tbits && logn <= P.logcap) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lds_fence();
                    distinct = wave_distinct_half(qlog, logn, mscr, P.count_tbits, max(P.id_bits, P.count_tbits - 1u), lane, bad);
                    lds_fence();
                }
                if (!bad) cmps += distinct;
                else {           // left to the host's exact recount (rg_search_wait)
                    cmps += logn;
                    if (lane == 0) P.ovf_list[atomicAdd(P.ovf_count, 1u)] = qi + P.qbase;
                }
                ++tot_left;
            }
            tot_n += (unsigned long long)pre + logn;
            tot_d += cmps;
        }
        if (VIS == 1 && qlog && P.count_mode == 1u && P.count_tbits && logn <= P.logcap && logn > 0 && !cmps_only && !build) {
            // the query is over: its log is complete (the tail stores above included) and the LDS from the merge scratch on
            // -- beam, log line, filter -- is free until the next query initialises it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_fence();
            bool bad = false;
            const uint32_t distinct = wave_distinct_half(qlog, logn, mscr, P.count_tbits, max(P.id_bits, P.count_tbits - 1u), lane, bad);
            if (!bad) {
                cmps = distinct;                 // the reference's cmps (:2397 counts every node once)
                logn_out = logn | kCountedBit;
                tot_n += logn;
                tot_d += distinct;
            }
            lds_fence();
        }
        if (lane == 0) {
            if (P.out_cmps) P.out_cmps[qi] = cmps;
            if (P.out_hops && !cmps_only) P.out_hops[qi] = hops;
            if (VIS == 1 && P.qlog_n) P.qlog_n[qi] = logn_out;
        }
        wave_sync();
    }
    if (LSET && tot_left && lane == 0) atomicAdd(P.lset_left, tot_left);
    if (LOGS && tot_n && lane == 0) {
        atomicAdd(&P.totals[0], tot_n);
        atomicAdd(&P.totals[1], tot_d);
    }
    if ((EXACT || ltags) && lane == 0) P.slot_epoch[blockIdx.x] = epoch;
}

// ------------------------------------------------------------------------------------------ launch plumbing
// what the host decided for one launch (rg_search.hip: plan_k1)
struct K1Launch {
    int R = 1;        // staging ring depth (passes of 4 rows in flight)
    int vis = 1;      // 0 = exact HBM visited words, 1 = LDS filter, 2 = exact words, look-ahead form, 3 = exact set in LDS
    int dimc = 0;     // compile-time dimension instantiation (0 = generic)
    bool bf = false;  // opt-in bf16 traversal
    int gf = 0;       // gather form of the register-staged instantiations: 0 = 16-byte loads + LDS bounce, 1 = compute layout
    uint32_t grid = 0;
    size_t lds = 0;
    int *occupancy = nullptr;   // non-null: do not launch, report the resident single-wave workgroups per CU of the kernel
};

template <bool L2, bool ELL, int R, int VIS, int DIMC, bool BF = false, int GF = 0>
static rg_status launch_search_d(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    auto kern = rg_search_kernel<L2, ELL, R, VIS, DIMC, BF, GF>;
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

// the look-ahead form exists for the register-staged instantiations only (k1_look_ok, rg_search.hip, asks for nothing else)
template <bool L2, bool ELL, int R>
static rg_status launch_search_look(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    if constexpr (ELL && R >= 2) {
        if (c.dimc == 200) return launch_search_d<L2, ELL, R, 2, 200>(P, c, s);
        if (c.dimc == 512 && R <= 4) return launch_search_d<L2, ELL, R, 2, 512>(P, c, s);
    }
    return set_error(RG_ERR_ARG, "internal: look-ahead form requested for a launch that has no such instantiation");
}

template <bool L2, bool ELL, int R>
static rg_status launch_search_t(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    if (c.vis == 2) return launch_search_look<L2, ELL, R>(P, c, s);
    return c.vis == 1 ? launch_search_v<L2, ELL, R, 1>(P, c, s) : launch_search_v<L2, ELL, R, 0>(P, c, s);
}

// every instantiation of one (metric, adjacency layout) family; one translation unit each (rg_search_inst_*.hip)
// compute-layout gather (GF = 1): d = 200 with four or eight register sets, d = 512 with two (k1_gf_ok, rg_search.hip)
template <bool L2, bool ELL, int R, int DIMC>
static rg_status launch_search_gf1(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    if (c.vis == 3) return launch_search_d<L2, ELL, R, 3, DIMC, false, 1>(P, c, s);      // exact set in LDS (these instantiations only)
    if (c.vis == 2) return launch_search_d<L2, ELL, R, 2, DIMC, false, 1>(P, c, s);
    return c.vis == 1 ? launch_search_d<L2, ELL, R, 1, DIMC, false, 1>(P, c, s) : launch_search_d<L2, ELL, R, 0, DIMC, false, 1>(P, c, s);
}

template <bool L2, bool ELL>
static rg_status launch_search_family(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    if constexpr (ELL) {
        if (c.gf == 1 && !c.bf) {
            if (c.dimc == 200 && c.R == 8) return launch_search_gf1<L2, ELL, 8, 200>(P, c, s);
            if (c.dimc == 200 && c.R == 4) return launch_search_gf1<L2, ELL, 4, 200>(P, c, s);
            if (c.dimc == 512 && c.R == 2) return launch_search_gf1<L2, ELL, 2, 512>(P, c, s);
            return set_error(RG_ERR_ARG, "internal: compute-layout gather requested for a launch that has no such instantiation");
        }
    }
    if constexpr (ELL) {   // eight register sets (32 rows in flight): d = 200 only, for launches with few resident queries
        if (c.R == 8 && c.dimc == 200 && !c.bf) {
            if (c.vis == 2) return launch_search_d<L2, ELL, 8, 2, 200>(P, c, s);
            return c.vis == 1 ? launch_search_d<L2, ELL, 8, 1, 200>(P, c, s) : launch_search_d<L2, ELL, 8, 0, 200>(P, c, s);
        }
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
