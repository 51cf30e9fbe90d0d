// rg_gt.hip -- brute-force k-NN ground truth on gfx950 (replaces the external DiskANN compute_groundtruth that
// RoarGraph's build consumes, README.md:62-75; consumer: LoadLearnBaseKNN, src/index_bipartite.cpp:2622-2642).
//
//   K2  rg_gt_kernel      fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain) scores of a block of
//                         MQ queries against the whole base shard, streamed 128 rows at a time, with a fused
//                         threshold filter + per-query candidate buffer + in-wave bitonic top-K
//   K2b rg_gt_rescore     exact (q-b)^2 re-score of the K survivors for L2 (the MFMA ranks by q.b - |b|^2/2)
//   K3  rg_gt_merge       merges per-shard sorted K-lists (multi-GPU: one list per rank) into the global top-K
//
// Work split: a workgroup (4 waves) OWNS a block of MQ queries for the whole pass, so the running thresholds and the
// candidate counters are LDS-local and nothing is shared between workgroups.  The query block is staged once in LDS,
// transposed to [k][query]; base rows stream through a double-buffered [k][row] LDS stage, prefetched through
// registers one k-chunk ahead.  All co-resident workgroups walk the base shard in the same order, so the shard is
// fetched from HBM about once per XCD per generation of workgroups and otherwise served by L2 / Infinity Cache.
#include <hip/hip_runtime.h>

#include <mutex>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "rg.h"
#include "rg_device.h"
#include "rg_internal.h"
#include "rg_mem.h"

namespace rg {

#define RG_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return set_error(RG_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr int kNB = 128;     // base rows per streamed tile

// order-preserving float -> uint (ascending)
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
// sort key, ascending = best first.  larger_first: score desc then id asc; else value asc then id asc.
__device__ __forceinline__ u64 make_key(float s, uint32_t id, bool larger_first) {
    const uint32_t o = f2ord(s + 0.0f);  // +0.0f folds -0 into +0
    return ((u64)(larger_first ? ~o : o) << 32) | id;
}
__device__ __forceinline__ float key_value(u64 k, bool larger_first) {
    const uint32_t hi = (uint32_t)(k >> 32);
    return ord2f(larger_first ? ~hi : hi);
}

// ascending bitonic sort of 64*ITEMS keys held as key[it] in lane `lane` (element index = it*64 + lane)
template <int ITEMS>
__device__ __forceinline__ void wave_sort(u64 (&key)[ITEMS], int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * ITEMS; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int dj = j >> 6;
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    const int pt = it ^ dj;
                    if (pt > it) {
                        const bool up = ((it * 64) & k) == 0;
                        const u64 a = key[it], b = key[pt];
                        const bool sw = (a > b) == up;
                        key[it] = sw ? b : a;
                        key[pt] = sw ? a : b;
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    const u64 mine = key[it];
                    const u64 other = (u64)__shfl_xor((long long)mine, j, 64);
                    const bool up = (((it * 64) + lane) & k) == 0;
                    const bool lower = (lane & j) == 0;
                    const u64 mn = mine < other ? mine : other, mx = mine < other ? other : mine;
                    key[it] = (lower == up) ? mn : mx;
                }
            }
        }
    }
}

struct GtParams {
    const float *base;
    uint32_t nb, bstride;
    const float *queries;
    uint32_t nq, qstride, dim;
    const float *bias;  // per base row added to q.b (null = 0): -|b|^2/2 for L2
    uint32_t K, id_base;
    uint32_t *out_ids;
    float *out_vals;    // the ranking value t = q.b + bias, best (largest) first
    u64 *cand;          // [grid][MQ][64*ITEMS]
    uint32_t *counter;
    uint32_t BK;
    uint32_t diag;  // ablation switches for profiling only (wrong results): 1 no base streaming, 2 no epilogue, 4 no per-chunk barrier; 16 (right results): mid-stream compaction by the bitonic sort instead of the selection
    // few query blocks for the chip: the base shard is also cut in nseg row segments, a work item is (query block, segment),
    // per-segment lists go to seg_ids / seg_vals [nseg][nq][K] and K3 merges them
    uint32_t nseg, seg_rows;
    uint32_t *seg_ids;
    float *seg_vals;
    // balanced form (round 4): the (query block x base tile) rectangle is cut into one equal stretch per resident workgroup;
    // a stretch that crosses a block boundary is two work items.  items[i] = (query block, first row, rows, list the piece's
    // top-K goes to), workgroup w runs items[item_first[w] .. item_first[w + 1]); null = the forms above (work counter)
    const uint4 *items;
    const uint32_t *item_first;
    // QUOTA THRESHOLDS (round 5).  A query block whose base rows are cut into p pieces (segments / balanced stretches) is searched by p
    // workgroups at once, each with the K-th best of ITS OWN rows as threshold -- the 100th best of a sixth of the base, where one
    // pass over the whole base would by then be filtering with the 100th best of everything seen.  Piece i therefore also publishes
    // u_i, the k_i-th best of its rows so far, k_i = ceil(K * rows_i / nb) (sum k_i >= K): every piece j holds at least k_j rows >=
    // u_j, so at least K rows of the base are >= min_j u_j -- a lower bound of the final K-th best that any piece may filter with.
    // quota_thr[(blk * nseg + piece) * MQB + query] (fp32 bits, -inf = nothing published yet); null = off (RG_GT_NOSHARE=1).
    uint32_t *quota_thr;
};

// the parameters of work item `item` = (query block, segment): the segment's rows, bias, id offset and output lists
__device__ __forceinline__ GtParams gt_segment(const GtParams &P0, uint32_t item, uint32_t &qblk, uint32_t &piece, uint32_t &npieces) {
    GtParams P = P0;
    piece = 0; npieces = 1;
    if (P0.items) {
        const uint4 it = P0.items[item];
        qblk = it.x;
        piece = it.w & 0xffffu; npieces = it.w >> 16;
        P.base = P0.base + (size_t)it.y * P0.bstride;
        P.nb = it.z;
        if (P0.bias) P.bias = P0.bias + it.y;
        P.id_base = P0.id_base + it.y;
        P.out_ids = P0.seg_ids + (size_t)(it.w & 0xffffu) * P0.nq * P0.K;      // (w = list | pieces of the block << 16)
        P.out_vals = P0.seg_vals + (size_t)(it.w & 0xffffu) * P0.nq * P0.K;
        return P;
    }
    qblk = item / P0.nseg;
    if (P0.nseg > 1) {
        const uint32_t seg = item % P0.nseg, r0 = seg * P0.seg_rows;
        piece = seg; npieces = P0.nseg;
        P.base = P0.base + (size_t)r0 * P0.bstride;
        P.nb = min(P0.seg_rows, P0.nb - r0);
        if (P0.bias) P.bias = P0.bias + r0;
        P.id_base = P0.id_base + r0;
        P.out_ids = P0.seg_ids + (size_t)seg * P0.nq * P0.K;
        P.out_vals = P0.seg_vals + (size_t)seg * P0.nq * P0.K;
    }
    return P;
}

// keep the best K of the query's candidate buffer, publish the new threshold
template <int ITEMS>
__device__ __forceinline__ void gt_compact(u64 *buf, uint32_t *cnt, float *thr, uint32_t K, int lane) {
    const uint32_t n = *cnt;
    u64 key[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint32_t e = it * 64 + lane;
        key[it] = e < n ? buf[e] : ~0ull;
    }
    wave_sort<ITEMS>(key, lane);
    const uint32_t keep = min(n, K);
    u64 kth = 0;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint32_t e = it * 64 + lane;
        if (e < keep) buf[e] = key[it];
        if ((uint32_t)it == (K - 1) / 64) kth = key[it];
    }
    kth = (u64)__shfl((long long)kth, (int)((K - 1) & 63), 64);
    if (lane == 0) {
        *cnt = keep;
        *thr = n >= K ? key_value(kth, true) : -__builtin_inff();
    }
}

// Round 5: the mid-stream compaction as a SELECTION instead of a sort.  Between the tiles a query's buffer only has to shed
// everything behind its K-th best and publish that value; the order of what stays does not matter until the item's final
// gt_compact.  One wave, ITEMS keys per lane: the K-th smallest high word (the ranking value) by bisection between the
// buffer's smallest and largest value with ballot counts -- a step is ITEMS compares + ballots on registers, no cross-lane
// data movement, against the 33 exchange stages of the 256-key bitonic network; ties of the K-th value are settled on the low
// word (the id) by a second bisection, in the rare buffers that have them; the survivors are packed with ballot prefix sums.
// Exactly the K smallest keys stay (keys are unique: the low word is the row id).
// quota thresholds of one query (GtParams::quota_thr): slots = the query's entry of piece 0, piece j's `pstride` words further
struct QuotaRef {
    uint32_t *slots = nullptr;     // null = off
    uint32_t npieces = 0, piece = 0, pstride = 0, quota = 0;
};
// The largest float below f (f itself for -inf).  The bound the pieces SHARE goes through this: the tile filter admits
// `score > threshold`, strictly -- right for a piece's OWN K-th value, whose holder is already in the buffer, but a row of another piece
// that ties with min_j u_j exactly (duplicated base rows in different segments) must still pass, or which of two equal rows survives
// would depend on when the bound arrived, and not on the id as the unsegmented path and K3 have it.
__device__ __forceinline__ float next_below(float f) {
    if (f == -__builtin_inff()) return f;
    f += 0.0f;                                                   // -0 -> +0
    if (f == 0.0f) return __uint_as_float(0x80000001u);          // the largest float below zero
    return ord2f(f2ord(f) - 1u);
}
// min over the pieces of what they published (-inf while some piece has published nothing); `mine` stands in for this piece's own slot.
// Any number of pieces (the balanced form allows 1024 / K of them: more than a wave's 64 lanes only at K <= 15)
__device__ __forceinline__ float quota_min(const QuotaRef &qr, float mine, int lane) {
    float t = __builtin_inff();
    for (uint32_t j = (uint32_t)lane; j < qr.npieces; j += 64u)
        t = fminf(t, j == qr.piece ? mine
                                   : __uint_as_float(__hip_atomic_load(qr.slots + (size_t)j * qr.pstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    for (int o = 32; o; o >>= 1) t = fminf(t, __shfl_xor(t, o, 64));
    return t;
}
template <int ITEMS>
__device__ __forceinline__ void gt_select(u64 *buf, uint32_t *cnt, float *thr, uint32_t K, int lane, uint32_t *q_slots, uint32_t q_pw, uint32_t q_quota,
                                          uint32_t q_pstride) {
    QuotaRef qr;      // (scalars at the call boundary: a struct by value goes through the stack)
    qr.slots = q_slots; qr.npieces = q_pw & 0xffffu; qr.piece = q_pw >> 16; qr.quota = q_quota; qr.pstride = q_pstride;
    const uint32_t n = *cnt;
    u64 key[ITEMS];
    uint32_t h[ITEMS];
    uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint32_t e = it * 64 + lane;
        key[it] = e < n ? buf[e] : ~0ull;
        h[it] = (uint32_t)(key[it] >> 32);
        if (e < n) { lo = min(lo, h[it]); hi = max(hi, h[it]); }
    }
    for (int o = 32; o; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
    }
    // smallest v in [l, r] with count(h <= v) >= k (needs n >= k; padding lanes are never counted)
    auto kth_value = [&](uint32_t k, uint32_t l, uint32_t r) __attribute__((always_inline)) -> uint32_t {
        while (l < r) {
            const uint32_t mid = l + ((r - l) >> 1);
            uint32_t c = 0;
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) c += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(h[it] <= mid && (uint32_t)(it * 64 + lane) < n));
            if (c >= k) r = mid; else l = mid + 1u;
        }
        return l;
    };
    float own = -__builtin_inff();
    uint32_t v = hi;
    if (n > K) {
        v = kth_value(K, lo, hi);
        uint32_t c_less = 0, c_eq = 0;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const bool live = (uint32_t)(it * 64 + lane) < n;
            c_less += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live && h[it] < v));
            c_eq += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live && h[it] == v));
        }
        uint32_t idmax = 0xffffffffu;       // keys with the K-th value stay up to this id
        const uint32_t r = K - c_less;      // how many of them stay (1 <= r <= c_eq)
        if (r < c_eq) {                     // ties at the K-th value: the r smallest ids
            uint32_t l2 = 0u, h2 = 0xffffffffu;
            while (l2 < h2) {
                const uint32_t mid = l2 + ((h2 - l2) >> 1);
                uint32_t c = 0;
#pragma unroll
                for (int it = 0; it < ITEMS; ++it)
                    c += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((uint32_t)(it * 64 + lane) < n && h[it] == v && (uint32_t)key[it] <= mid));
                if (c >= r) h2 = mid; else l2 = mid + 1u;
            }
            idmax = l2;
        }
        uint32_t base = 0;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const bool keep = (uint32_t)(it * 64 + lane) < n && (h[it] < v || (h[it] == v && (uint32_t)key[it] <= idmax));
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            if (keep) buf[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key[it];
            base += (uint32_t)__popcll(m);
        }
        own = ord2f(~v);                    // (make_key stores ~ord of the score: larger scores first)
    } else if (n == K) {
        own = ord2f(~hi);
    }
    float t = own;
    if (qr.slots) {
        // this piece's k_i-th best so far (it only grows), published; then the bound every piece may use
        float mine = -__builtin_inff();
        if (n >= qr.quota) {
            mine = ord2f(~kth_value(qr.quota, lo, v));
            if (lane == 0) __hip_atomic_store(qr.slots + (size_t)qr.piece * qr.pstride, __float_as_uint(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        t = fmaxf(own, next_below(quota_min(qr, mine, lane)));
    }
    if (lane == 0) {
        if (n > K) *cnt = K;
        *thr = fmaxf(*thr, t);              // (a threshold never loosens: the bound of an earlier event may have been the tighter one)
    }
}
template <int ITEMS>
__device__ __attribute__((noinline)) void gt_select_call(u64 *buf, uint32_t *cnt, float *thr, uint32_t K, int lane, uint32_t *q_slots, uint32_t q_pw,
                                                         uint32_t q_quota, uint32_t q_pstride) {
    gt_select<ITEMS>(buf, cnt, thr, K, lane, q_slots, q_pw, q_quota, q_pstride);
}

typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;

// out-of-line form for call sites inside register-saturated loops (the rare path pays the call, the loop keeps its registers)
template <int ITEMS>
__device__ __attribute__((noinline)) void gt_compact_call(u64 *buf, uint32_t *cnt, float *thr, uint32_t K, int lane) {
    gt_compact<ITEMS>(buf, cnt, thr, K, lane);
}

// LDS layout of both operands: k-quads, [k/4][row][4 floats] -- one 16-byte slot per (k-quad, row).
//  * the base chunk arrives by LDS-DMA (global_load_lds_dwordx4): a wave instruction fills 64 consecutive slots
//    (64 rows of one k-quad), no VGPR staging and no transposing ds_write pass;
//  * an MFMA operand fetch is one conflict-free ds_read_b64 per lane (rows are consecutive slots) that serves TWO
//    v_mfma_f32_32x32x2_f32: lanes 0-31 read elements (0,1) of the quad, lanes 32-63 elements (2,3); the first MFMA
//    multiplies the k pair (0,2), the second (1,3) (operand lane l holds k-slot l >> 5 of the pair).
template <int MQ, int ITEMS>
__global__ void __launch_bounds__(512) rg_gt_kernel(GtParams P0) {
    constexpr int C = 64 * ITEMS;
    constexpr int TM = MQ / 64;              // 32-row query tiles per wave (2 for MQ=128, 1 for MQ=64)
    constexpr int NW = 8;                    // waves per workgroup: 2 (query axis) x 4 (base axis), two per SIMD
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 3, wm = w >> 2;
    const int qoff = 32 * TM * wm, boff = 32 * wn;
    const uint32_t BK = P0.BK, dim = P0.dim;
    const uint32_t kq_chunk = BK / 4;                               // k-quads per chunk

    float4 *Qq = reinterpret_cast<float4 *>(smem);                  // [dim/4][MQ]
    float4 *Bq = Qq + (size_t)(dim / 4) * MQ;                       // [2][BK/4][128]
    float *thr = reinterpret_cast<float *>(Bq + 2 * (size_t)kq_chunk * kNB);   // [MQ]
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + MQ);         // [MQ]
    uint32_t *flag = cnt + MQ;                                      // [4]
    u64 *cand = P0.cand + (size_t)blockIdx.x * MQ * C;

    const uint32_t nkc = dim / BK;
    const uint32_t ninstr = kq_chunk * (kNB / 64);                  // LDS-DMA wave instructions per chunk (2 per k-quad)
    const bool hi = lane >= 32;

    uint32_t cursor = P0.items ? P0.item_first[blockIdx.x] : 0u;
    const uint32_t cursor_end = P0.items ? P0.item_first[blockIdx.x + 1] : 0u;
    for (;;) {
        uint32_t item;
        if (P0.items) {                       // balanced form: this workgroup's own stretch
            if (cursor >= cursor_end) break;
            item = cursor++;
            __syncthreads();
        } else {
            if (tid == 0) flag[1] = atomicAdd(P0.counter, 1u);
            __syncthreads();
            item = flag[1];
            __syncthreads();
        }
        uint32_t blk;
        uint32_t piece, npieces;
        const GtParams P = gt_segment(P0, item, blk, piece, npieces);
        if ((uint64_t)blk * MQ >= P.nq) break;
        const uint32_t ntiles = (P.nb + kNB - 1) / kNB;
        const uint32_t q0 = blk * MQ;
        // stage the query block: Qq[k/4][q] (16-byte stores, consecutive rows -> consecutive slots)
        for (uint32_t idx = tid; idx < (uint32_t)MQ * (dim / 4); idx += 512) {
            const uint32_t row = idx % MQ, f4 = idx / MQ;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q0 + row < P.nq) v = *reinterpret_cast<const float4 *>(P.queries + (size_t)(q0 + row) * P.qstride + 4 * f4);
            Qq[(size_t)f4 * MQ + row] = v;
        }
        for (int i = tid; i < MQ; i += 512) { thr[i] = -__builtin_inff(); cnt[i] = 0; }
        if (tid == 0) flag[0] = 0;

        // LDS-DMA of base chunk c into buffer `buf`: instruction j covers k-quad j/2, rows 64*(j&1) .. +63
        auto stream_chunk = [&](uint32_t c, uint32_t buf) {
            const uint32_t tile = c / nkc, k0 = (c % nkc) * BK;
            float4 *dst = Bq + (size_t)buf * kq_chunk * kNB;
            for (uint32_t j = (uint32_t)w; j < ninstr; j += NW) {
                const uint32_t kq = j >> 1, row = 64u * (j & 1u) + (uint32_t)lane;
                const uint32_t gr = min(tile * kNB + row, P.nb - 1u);   // clamp: scores of rows past the end are ignored
                const float *src = P.base + (size_t)gr * P.bstride + k0 + 4 * kq;
                // issued through inline asm on purpose: hipcc treats the builtin as an LDS write it cannot disambiguate
                // and parks an s_waitcnt vmcnt(0) in front of the very next ds_read (of the OTHER buffer), which would
                // serialise the stream with the MFMAs.  The DMA is waited for explicitly before the chunk's barrier.
                const uint32_t lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t *)(dst + (size_t)j * 64));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                             :: "s"(lds_addr), "v"(src) : "memory");
            }
        };

        f32x16 acc[TM];
        auto init_acc = [&](uint32_t tile) {
            const uint32_t id = tile * kNB + boff + (lane & 31);
            const float b = (P.bias && id < P.nb) ? P.bias[id] : 0.0f;
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = b;
        };

        const uint32_t nchunks = ntiles * nkc;
        stream_chunk(0, 0);
        init_acc(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (uint32_t c = 0; c < nchunks; ++c) {
            const uint32_t buf = c & 1u;
            if (c + 1 < nchunks && !(P.diag & 1u)) stream_chunk(c + 1, buf ^ 1u);
            // operand fragments: the low half-wave takes elements (0,1) of a k-quad, the high half-wave (2,3) -- the two
            // MFMAs of a quad multiply the k pairs (0,2) and (1,3), so one ds_read_b64 per operand feeds both with no
            // lane select (the order of the k sum is free here)
            const float2 *bq = reinterpret_cast<const float2 *>(Bq + (size_t)buf * kq_chunk * kNB + boff + (lane & 31)) + (hi ? 1 : 0);
            const float2 *qq = reinterpret_cast<const float2 *>(Qq + (size_t)((c % nkc) * kq_chunk) * MQ + qoff + (lane & 31)) + (hi ? 1 : 0);
            // two k-quads per step: operands of step s+1 are read from LDS while the 4*TM MFMAs of step s issue
            float2 a0[TM], b0, a1[TM], b1;
            auto fetch = [&](uint32_t kq, float2 (&a)[TM], float2 &b) {
                b = bq[2 * ((size_t)kq * kNB)];
#pragma unroll
                for (int m = 0; m < TM; ++m) a[m] = qq[2 * ((size_t)kq * MQ + 32 * m)];
            };
            auto mfma_quad = [&](const float2 (&a)[TM], const float2 &b) {
#pragma unroll
                for (int m = 0; m < TM; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, b.x, acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < TM; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, b.y, acc[m], 0, 0, 0);
            };
            fetch(0, a0, b0);
            for (uint32_t kq = 0; kq < kq_chunk; kq += 2) {
                if (kq + 1 < kq_chunk) fetch(kq + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_quad(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (kq + 2 < kq_chunk) fetch(kq + 2, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (kq + 1 < kq_chunk) mfma_quad(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((c + 1) % nkc == 0 && !(P.diag & 2u)) {
                // tile finished: threshold filter, survivors -> candidate buffers
                const uint32_t tile = c / nkc;
                const uint32_t id = tile * kNB + boff + (lane & 31);
                // all threshold reads first (independent LDS loads), then the rare survivors one by one
                uint32_t win = 0;
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qi = qoff + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        win |= (acc[m][r] > thr[qi] ? 1u : 0u) << (16 * m + r);
                    }
                if (id >= P.nb) win = 0;
                if (win) {
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (win & (1u << (16 * m + r))) {
                                const int qi = qoff + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                                const uint32_t slot = atomicAdd(&cnt[qi], 1u);
                                cand[(size_t)qi * C + slot] = make_key(acc[m][r], id, true);
                                if (slot + 1 + kNB > (uint32_t)C) flag[0] = 1;
                            }
                }
                if (tile + 1 < ntiles) init_acc(tile + 1);
                __syncthreads();
                if (flag[0]) {
                    for (int qi = w; qi < MQ; qi += NW)
                        if (cnt[qi] + kNB > (uint32_t)C) gt_compact<ITEMS>(cand + (size_t)qi * C, &cnt[qi], &thr[qi], P.K, lane);
                    __syncthreads();
                    if (tid == 0) flag[0] = 0;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next chunk's DMA has landed
            if (!(P.diag & 4u)) __syncthreads();
        }
        // final selection + output
        for (int qi = w; qi < MQ; qi += NW) {
            gt_compact<ITEMS>(cand + (size_t)qi * C, &cnt[qi], &thr[qi], P.K, lane);
            const uint32_t q = q0 + qi;
            if (q < P.nq) {
                for (uint32_t e = lane; e < P.K; e += 64) {
                    const u64 k = cand[(size_t)qi * C + e];
                    P.out_ids[(size_t)q * P.K + e] = (uint32_t)k + P.id_base;
                    P.out_vals[(size_t)q * P.K + e] = key_value(k, true);
                }
            }
        }
        __syncthreads();
    }
}

// one LDS-DMA wave instruction: 64 lanes x 16 B from src + OFF_BYTES to LDS.  The instruction's immediate offset is
// added to BOTH the global and the LDS address (LDS address = M0 + offset + 16 * lane), hence M0 = lds_addr - OFF_BYTES.
// Issued through inline asm on purpose (see rg_gt_kernel::stream_chunk).
template <int OFF_BYTES>
__device__ __forceinline__ void glds16(uint32_t lds_addr, const float *src) {
    static_assert(OFF_BYTES >= 0 && OFF_BYTES < 4096, "immediate offset range");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%2"
                 :: "s"(lds_addr - (uint32_t)OFF_BYTES), "v"(src), "n"(OFF_BYTES) : "memory");
}
// N instructions: instruction i moves k-quad kq0 + 2i of the chunk (src advances 32 B, LDS 4 KiB = two 64-row halves)
template <int I, int N>
__device__ __forceinline__ void glds_run(uint32_t lds_addr, const float *src) {
    if constexpr (I < N) {
        glds16<32 * I>(lds_addr + 4096u * I, src);
        glds_run<I + 1, N>(lds_addr, src);
    }
}

// K2-RS: "queries stationary in registers" form of K2 for the dimensions of the BASELINE configs (200, 512).
// A wave keeps the MFMA A-operands of its 32*TMW queries for ALL k in VGPRs (DIM/2 registers per 32-query tile: lane l
// holds, for k-quad kq, Q[q = l&31][4kq + 2*(l>>5)] and Q[q][4kq + 1 + 2*(l>>5)] -- the two MFMAs of a quad take the k
// pairs (0,2) and (1,3) so that each half-wave needs two ADJACENT floats of the base row's quad), owns all four 32-column tiles of the streamed base tile, and therefore issues
// 8*TMW MFMAs per k-pair with only the B fragments coming from LDS.  One wave per SIMD (4 waves = one workgroup per CU,
// 128*TMW queries); LDS holds just the double-buffered base chunk, barriers are BK/2 * 8*TMW MFMAs apart (10k cycles at
// d=200) and the base stream is shared by twice as many queries as in the LDS-resident form.  Everything in k is
// unrolled at compile time (register arrays need static indices), hence the DIM template parameter.
template <int DIM, int BK, int TMW, int ITEMS, int WPS, bool PROF = false>
__global__ void __launch_bounds__(256, WPS) rg_gt_rs_kernel(GtParams P0) {
    // PROF (RG_GT_PROF=1, d = 200 only): per-wave sums of s_memtime ticks spent in the tile filter, in
    // its rare path and in compactions, with their counts, added to P0.counter[4 ..] (u64) by wave 0 of every workgroup
    unsigned long long pf_filter = 0, pf_slow = 0, pf_compact = 0, pf_item = 0;
    uint32_t pf_tiles = 0, pf_nslow = 0, pf_nev = 0, pf_ncand = 0;
    constexpr int C = 64 * ITEMS;
    constexpr int MQB = 128 * TMW;            // queries per workgroup
    constexpr int NKC = DIM / BK;             // k-chunks per base tile
    constexpr int KQC = BK / 4;               // k-quads per chunk
    static_assert(DIM % BK == 0 && BK % 4 == 0, "chunking");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qoff = 32 * TMW * w;
    const bool hi = lane >= 32;

    float4 *Bq = reinterpret_cast<float4 *>(smem);                               // [2][KQC][128]
    float *thr = reinterpret_cast<float *>(Bq + 2 * (size_t)KQC * kNB);          // [MQB]
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + MQB);                     // [MQB]
    uint32_t *flag = cnt + MQB;                                                  // [4]
    float *bias_l = reinterpret_cast<float *>(flag + 4);                         // [2][128] -|b|^2/2 of the tile's rows (L2)
    u64 *cand = P0.cand + (size_t)blockIdx.x * MQB * C;

    uint32_t cursor = P0.items ? P0.item_first[blockIdx.x] : 0u;
    const uint32_t cursor_end = P0.items ? P0.item_first[blockIdx.x + 1] : 0u;
    for (;;) {
        uint32_t item;
        if (P0.items) {                       // balanced form: this workgroup's own stretch
            if (cursor >= cursor_end) break;
            item = cursor++;
            __syncthreads();
        } else {
            if (tid == 0) flag[1] = atomicAdd(P0.counter, 1u);
            __syncthreads();
            item = flag[1];
            __syncthreads();
        }
        uint32_t blk;
        uint32_t piece, npieces;
        const GtParams P = gt_segment(P0, item, blk, piece, npieces);
        if ((uint64_t)blk * MQB >= P.nq) break;
        const uint32_t ntiles = (P.nb + kNB - 1) / kNB;
        const uint32_t q0 = blk * MQB;
        // quota thresholds of this item's piece (GtParams::quota_thr): k_i = ceil(K * rows of the piece / rows of the shard).  Parked in
        // LDS (flag[2], flag[3]) and read back in the rare event path: carried in registers through the tile loop they spill.
        if (tid == 0) {
            const bool on = P0.quota_thr && npieces > 1;
            flag[2] = on ? (npieces | (piece << 16)) : 0u;
            flag[3] = (uint32_t)(((unsigned long long)P.K * P.nb + P0.nb - 1u) / P0.nb);
        }
        const unsigned long long pf_t0 = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
        // A operands of this wave's queries, all k, into registers
        float areg[DIM / 2][TMW];
#pragma unroll
        for (int m = 0; m < TMW; ++m) {
            const uint32_t q = q0 + qoff + 32 * m + (lane & 31);
            const float *qrow = P.queries + (size_t)min(q, P.nq - 1u) * P.qstride + (hi ? 2 : 0);
            const float scale = q < P.nq ? 1.0f : 0.0f;
#pragma unroll
            for (int kk = 0; kk < DIM / 2; ++kk) areg[kk][m] = qrow[4 * (kk >> 1) + (kk & 1)] * scale;
        }
        for (int i = tid; i < MQB; i += 256) { thr[i] = -__builtin_inff(); cnt[i] = 0; }
        if (tid == 0) flag[0] = 0;
        // thresholds of this lane's query rows: kept in registers when the budget allows (TMW == 1), else read from LDS
        constexpr bool kThrRegs = TMW == 1 && WPS == 1;   // two workgroups per CU run at the 256-VGPR limit: thresholds from LDS
        float thr_r[kThrRegs ? TMW : 1][16];
        if (kThrRegs) {
#pragma unroll
            for (int r = 0; r < 16; ++r) thr_r[0][r] = -__builtin_inff();
        }

        // LDS-DMA of a base chunk.  Wave w always moves row half (w & 1) and the k-quads (w >> 1) + 2i, so a lane's
        // source is one row pointer per tile plus a per-chunk constant and an immediate: one 64-bit add per chunk and two
        // scalar ops per instruction.
        static_assert(KQC % 2 == 0, "k-quads per chunk");
        const uint32_t lds_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t *)Bq) +
                               (2u * ((uint32_t)w >> 1) + ((uint32_t)w & 1u)) * 1024u;
        auto row_ptr = [&](uint32_t tile) {
            // (the lane number is made here, as in stream_bias below: carried into the tile loop it is spilled in the instantiations with
            // larger candidate buffers, and the reload of a spill waits for vmcnt(0) -- for every DMA in flight, once per tile)
            uint32_t ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const uint32_t gr = min(tile * kNB + 64u * ((uint32_t)w & 1u) + ln, P.nb - 1u);   // clamp: rows past the end are ignored
            return P.base + (size_t)gr * P.bstride + 4u * ((uint32_t)w >> 1);
        };
        auto stream_chunk = [&](auto cc, const float *rowp, uint32_t buf) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            glds_run<0, KQC / 2>(lds_w + buf * (uint32_t)(KQC * kNB * 16), rowp + c * BK);
        };
        const float *rowp = row_ptr(0);

        f32x16 acc[TMW][4];
        uint32_t step = 0;   // running chunk counter: LDS buffer = step & 1
        {
            // The chunk's closing barrier sits BEFORE its last k-quad.  After it a wave issues the DMA
            // of the chunk after next into the buffer just read, prefetches the first fragments of the next chunk, and
            // only then multiplies its last quad: DMA issue, LDS latency and barrier skew hide behind 8 MFMAs instead
            // of opening every chunk.  The compaction check rides on the barrier of a tile's first chunk; the L2 bias of
            // a tile arrives with its first chunk (one more DMA instruction) instead of a per-tile global load.
            static_assert(NKC >= 2 && NKC <= 8 && KQC % 2 == 0 && KQC / 2 <= 16, "chunking");
            float2 b0[4], b1[4];
            const uint32_t lds_bias = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t *)bias_l);
            auto stream_bias = [&](uint32_t tile, uint32_t slot) __attribute__((always_inline)) {
                if (P.bias && w < 2) {   // waves 0/1: rows 0-63 / 64-127 of the tile
                    // (the lane number is made HERE: a value carried into the loop is spilled at this register pressure, and the
                    // reload of a spill waits for vmcnt(0) -- for every DMA in flight, once per tile)
                    uint32_t ln;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                    const uint32_t id = min(tile * kNB + 64u * (uint32_t)w + ln, P.nb - 1u);
                    const float *src = P.bias + id;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
                                 :: "s"(lds_bias + slot * 512u + 256u * (uint32_t)w), "v"(src) : "memory");
                }
            };
            const float *rowp_next = rowp;
            stream_chunk(std::integral_constant<int, 0>{}, rowp, 0);
            stream_bias(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            stream_chunk(std::integral_constant<int, 1>{}, rowp, 1);
            {
                const float2 *bq = reinterpret_cast<const float2 *>(Bq + (lane & 31)) + (hi ? 1 : 0);
#pragma unroll
                for (int n = 0; n < 4; ++n) b0[n] = bq[2 * (32 * n)];
            }
            for (uint32_t tile = 0; tile < ntiles; ++tile) {
                // accumulators of a tile start as the rows' bias (L2: -|b|^2/2).  Without a bias (IP) nothing is written here: the first MFMA of
                // the tile takes the constant 0 as its C operand instead (round 6) -- the 64 v_mov of the initialisation sat between the filter
                // of one tile and the first MFMA of the next, with nothing in the matrix pipe
                const bool has_bias = P.bias != nullptr;
                if (has_bias) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const float b = bias_l[(tile & 1u) * 128u + 32 * n + (lane & 31)];
#pragma unroll
                        for (int m = 0; m < TMW; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[m][n][r] = b;
                    }
                }
                auto chunk = [&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    const uint32_t buf = step & 1u;
                    const float2 *bq = reinterpret_cast<const float2 *>(Bq + (size_t)buf * KQC * kNB + (lane & 31)) + (hi ? 1 : 0);
                    const float2 *bqn = reinterpret_cast<const float2 *>(Bq + (size_t)(buf ^ 1u) * KQC * kNB + (lane & 31)) + (hi ? 1 : 0);
#pragma unroll
                    for (int kq = 0; kq < KQC; ++kq) {
                        float2 (&bc)[4] = (kq & 1) ? b1 : b0;
                        float2 (&bn)[4] = (kq & 1) ? b0 : b1;
                        if (kq + 1 < KQC) {
#pragma unroll
                            for (int n = 0; n < 4; ++n) bn[n] = bq[2 * ((kq + 1) * kNB + 32 * n)];
                        } else {
                            // every LDS read of this buffer has landed (the last quad's fragments are in registers), the
                            // next chunk's DMA (and this wave's candidate stores) too: close the chunk
                            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                            __syncthreads();
                            if (c == 0 && flag[0]) {   // set by the previous tile's filter
                                const unsigned long long pf_c0 = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
                                if (PROF) ++pf_nev;
                                for (int qi = w; qi < MQB; qi += 4)
                                    if (cnt[qi] + kNB > (uint32_t)C) {
                                        // selection (round 5); RG_GT_DIAG=16: the bitonic sort of rounds 1 - 4 (same lists, A/B)
                                        bool sorted_form = false;
                                        if constexpr ((ITEMS & (ITEMS - 1)) == 0) {
                                            if (P.diag & 16u) { gt_compact_call<ITEMS>(cand + (size_t)qi * C, &cnt[qi], &thr[qi], P.K, lane); sorted_form = true; }
                                        }
                                        if (!sorted_form) {
                                            const uint32_t pw = (uint32_t)__builtin_amdgcn_readfirstlane((int)flag[2]);
                                            uint32_t *sl = pw ? P0.quota_thr + (size_t)(q0 / MQB) * P0.nseg * MQB + qi : nullptr;
                                            gt_select_call<ITEMS>(cand + (size_t)qi * C, &cnt[qi], &thr[qi], P.K, lane, sl, pw,
                                                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)flag[3]), (uint32_t)MQB);
                                        }
                                    }
                                {   // every query of this wave takes up what the other pieces have published since
                                    const uint32_t pw = (uint32_t)__builtin_amdgcn_readfirstlane((int)flag[2]);
                                    if (pw && lane < 32) {
                                        const int qi = w + 4 * lane;
                                        const uint32_t *sl = P0.quota_thr + (size_t)(q0 / MQB) * P0.nseg * MQB + qi;
                                        float t = __builtin_inff();
                                        for (uint32_t j = 0; j < (pw & 0xffffu); ++j)
                                            t = fminf(t, __uint_as_float(__hip_atomic_load(sl + (size_t)j * MQB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
                                        thr[qi] = fmaxf(thr[qi], next_below(t));      // (strictly below the bound: its ties must pass)
                                    }
                                }
                                __syncthreads();
                                if (tid == 0) flag[0] = 0;
                                if (kThrRegs) {
#pragma unroll
                                    for (int r = 0; r < 16; ++r) thr_r[0][r] = thr[qoff + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
                                }
                                __syncthreads();
                                if (PROF) pf_compact += __builtin_amdgcn_s_memtime() - pf_c0;
                            }
                            // chunk after next -> the buffer this chunk was read from.  Its DMA instructions and the first
                            // fragments of the next chunk are issued BETWEEN the last quad's MFMAs (one per MFMA), so that
                            // each goes out in the shadow of a running MFMA instead of in front of the first one
                            const float *dsrc = nullptr;
                            if (!(P.diag & 1u)) {
                                if constexpr (c + 2 < NKC) {
                                    dsrc = rowp + (c + 2) * BK;
                                } else if (tile + 1 < ntiles) {
                                    if constexpr (c + 2 == NKC) {
                                        rowp_next = row_ptr(tile + 1);
                                        stream_bias(tile + 1, (tile + 1) & 1u);
                                    }
                                    dsrc = rowp_next + (c + 2 - NKC) * BK;
                                }
                            }
                            const uint32_t dlds = lds_w + buf * (uint32_t)(KQC * kNB * 16);
                            const bool next_frags = c + 1 < NKC || tile + 1 < ntiles;
                            const int kkt = 2 * (c * KQC + kq);
                            auto tail_step = [&](auto jj) __attribute__((always_inline)) {
                                constexpr int j = decltype(jj)::value;
                                constexpr int n = j & 3;
#pragma unroll
                                for (int m = 0; m < TMW; ++m)
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[kkt + (j >> 2)][m], (j < 4) ? bc[n].x : bc[n].y,
                                                                                     acc[m][n], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (j < KQC / 2) {
                                    if (dsrc) glds16<32 * j>(dlds + 4096u * j, dsrc);
                                }
                                if constexpr (j + 8 < KQC / 2) {      // (chunks of more than 16 k-quads: two DMA instructions per step)
                                    if (dsrc) glds16<32 * (j + 8)>(dlds + 4096u * (j + 8), dsrc);
                                }
                                if constexpr (j >= 4) {   // the .x halves of bc are consumed: refill that register set's slot n
                                    if (next_frags) bn[n] = bqn[2 * (32 * n)];
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            };
                            __builtin_amdgcn_sched_barrier(0);
                            tail_step(std::integral_constant<int, 0>{}); tail_step(std::integral_constant<int, 1>{});
                            tail_step(std::integral_constant<int, 2>{}); tail_step(std::integral_constant<int, 3>{});
                            tail_step(std::integral_constant<int, 4>{}); tail_step(std::integral_constant<int, 5>{});
                            tail_step(std::integral_constant<int, 6>{}); tail_step(std::integral_constant<int, 7>{});
                            continue;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const int kk = 2 * (c * KQC + kq);
                        if (c == 0 && kq == 0 && !has_bias) {      // first k pair of the tile, no bias: C = 0
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int n = 0; n < 4; ++n)
#pragma unroll
                                for (int m = 0; m < TMW; ++m)
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[kk][m], bc[n].x, zero, 0, 0, 0);
                        } else {
#pragma unroll
                            for (int n = 0; n < 4; ++n)
#pragma unroll
                                for (int m = 0; m < TMW; ++m)
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[kk][m], bc[n].x, acc[m][n], 0, 0, 0);
                        }
#pragma unroll
                        for (int n = 0; n < 4; ++n)
#pragma unroll
                            for (int m = 0; m < TMW; ++m)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[kk + 1][m], bc[n].y, acc[m][n], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ++step;
                    if (c + 1 == NKC && !(P.diag & 2u)) {
                        // tile finished: threshold filter (no barrier here: the compaction check is at the next barrier)
                        // (two instructions per accumulator row for the maximum -- fmaxf would canonicalise its inputs first -- and
                        // the wave's verdict as an OR of ballots on the scalar unit: as "any_win |= mx > t" the compiler builds a
                        // 16-bit mask per lane, a hundred vector instructions per tile)
                        const unsigned long long pf_f0 = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
                        if (PROF) ++pf_tiles;
                        // (round 5, measured and removed: a quick reject in front of the sixteen compares -- one maximum tree against the smallest
                        // threshold of the lane's rows -- level at 10,000 - 100,000 queries and 1 % SLOWER at 65,536: gt_ab_box25_quick_reject.jsonl)
                        uint64_t any_win = 0;
                        float mx[TMW][16];
#pragma unroll
                        for (int m = 0; m < TMW; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float t = kThrRegs ? thr_r[kThrRegs ? m : 0][r]
                                                         : thr[qoff + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
                                float m3;
                                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(acc[m][0][r]), "v"(acc[m][1][r]), "v"(acc[m][2][r]));
                                asm("v_max_f32 %0, %1, %2" : "=v"(mx[m][r]) : "v"(m3), "v"(acc[m][3][r]));
                                any_win |= __builtin_amdgcn_ballot_w64(mx[m][r] > t);
                            }
                        if (any_win) {
                            const unsigned long long pf_s0 = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
                            if (PROF) ++pf_nslow;
#pragma unroll
                            for (int m = 0; m < TMW; ++m)
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int qi = qoff + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                                    const float t = kThrRegs ? thr_r[kThrRegs ? m : 0][r] : thr[qi];
                                    if (mx[m][r] > t) {
#pragma unroll
                                        for (int n = 0; n < 4; ++n) {
                                            const uint32_t id = tile * kNB + 32 * n + (lane & 31);
                                            if (acc[m][n][r] > t && id < P.nb && !(P.diag & 8u)) {
                                                const uint32_t slot = atomicAdd(&cnt[qi], 1u);
                                                // the buffer's address is made HERE from the kernel argument: hoisted out of the loop it
                                                // is spilled, and the reload of a spill waits for vmcnt(0) -- i.e. for the DMA of the next
                                                // chunk and for the previous candidate's store, on every candidate
                                                __attribute__((address_space(1))) u64 *cb = (__attribute__((address_space(1))) u64 *)P0.cand;
                                                asm volatile("" : "+s"(cb));
                                                // (and the buffer's index from a lane number made here: with 384-key buffers the product
                                                // (block * MQB + query) * C is otherwise carried through the tile loop in a spilled register)
                                                uint32_t ln;
                                                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                                                const uint32_t qi2 = (uint32_t)(qoff + 32 * m + (r & 3) + 8 * (r >> 2)) + 4u * (ln >> 5);
                                                cb[((uint32_t)blockIdx.x * MQB + qi2) * (uint32_t)C + slot] = make_key(acc[m][n][r], id, true);
                                                if (slot + 1 + kNB > (uint32_t)C) flag[0] = 1;
                                                if (PROF) pf_ncand += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(true));
                                            }
                                        }
                                    }
                                }
                            if (PROF) pf_slow += __builtin_amdgcn_s_memtime() - pf_s0;
                        }
                        if (PROF) pf_filter += __builtin_amdgcn_s_memtime() - pf_f0;
                    }
                    if (c + 1 == NKC) rowp = rowp_next;
                };
                if constexpr (0 < NKC) chunk(std::integral_constant<int, 0>{});
                if constexpr (1 < NKC) chunk(std::integral_constant<int, 1>{});
                if constexpr (2 < NKC) chunk(std::integral_constant<int, 2>{});
                if constexpr (3 < NKC) chunk(std::integral_constant<int, 3>{});
                if constexpr (4 < NKC) chunk(std::integral_constant<int, 4>{});
                if constexpr (5 < NKC) chunk(std::integral_constant<int, 5>{});
                if constexpr (6 < NKC) chunk(std::integral_constant<int, 6>{});
                if constexpr (7 < NKC) chunk(std::integral_constant<int, 7>{});
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last tile's candidate stores
            __syncthreads();
        }
        const unsigned long long pf_t1 = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
        // final selection + output
        for (int qi = w; qi < MQB; qi += 4) {
            const uint32_t kept = min(cnt[qi], P.K);     // (a piece that filtered with the other pieces' bound may end with fewer than K)
            if constexpr (ITEMS > 4) {     // larger buffers: shed to K with the selection, then the 256-key network orders what is left
                if (cnt[qi] > P.K) gt_select_call<ITEMS>(cand + (size_t)qi * C, &cnt[qi], &thr[qi], P.K, lane, nullptr, 0u, 0u, 0u);
                gt_compact<4>(cand + (size_t)qi * C, &cnt[qi], &thr[qi], P.K, lane);
            } else gt_compact<ITEMS>(cand + (size_t)qi * C, &cnt[qi], &thr[qi], P.K, lane);
            const uint32_t q = q0 + qi;
            if (q < P.nq) {
                for (uint32_t e = lane; e < P.K; e += 64) {
                    const u64 k = cand[(size_t)qi * C + e];
                    P.out_ids[(size_t)q * P.K + e] = e < kept ? (uint32_t)k + P.id_base : 0xffffffffu;      // padding ranks last in K3
                    P.out_vals[(size_t)q * P.K + e] = e < kept ? key_value(k, true) : -__builtin_inff();
                }
            }
        }
        __syncthreads();
        if (PROF) pf_item += pf_t1 - pf_t0;
    }
    if (PROF && w == 0 && lane == 0) {
        unsigned long long *pf = reinterpret_cast<unsigned long long *>(P0.counter + 4);
        atomicAdd(pf + 0, pf_item); atomicAdd(pf + 1, pf_filter); atomicAdd(pf + 2, pf_slow); atomicAdd(pf + 3, pf_compact);
        atomicAdd(pf + 4, (unsigned long long)pf_tiles); atomicAdd(pf + 5, (unsigned long long)pf_nslow);
        atomicAdd(pf + 6, (unsigned long long)pf_nev); atomicAdd(pf + 7, (unsigned long long)pf_ncand);
    }
}

// the item table of the balanced form, made on the device (one thread: <= 2 * slots + nblocks items) so that the launch
// needs no host-to-device copy on the compute stream; the host runs the same loop to size and validate it
__host__ __device__ inline uint64_t gt_cut(uint32_t w, uint32_t slots, uint64_t per, uint64_t total, uint64_t tpb, uint64_t snap) {
    if (w >= slots) return total;
    uint64_t c = (uint64_t)w * per;
    if (c > total) c = total;
    const uint64_t r = c % tpb;
    if (r != 0 && r < snap) c -= r;
    else if (r != 0 && tpb - r < snap) c += tpb - r;
    return c < total ? c : total;
}
__global__ void rg_gt_items_kernel(uint32_t slots, uint64_t per, uint64_t total, uint64_t tpb, uint64_t snap, uint32_t nb, uint4 *items, uint32_t *first) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t n = 0, cur_blk = 0xffffffffu, cur_list = 0, blk_first = 0;
    auto close_block = [&]() { for (uint32_t i = blk_first; i < n; ++i) items[i].w |= cur_list << 16; };     // w = list | pieces of the block << 16
    for (uint32_t w = 0; w < slots; ++w) {
        first[w] = n;
        const uint64_t hi = gt_cut(w + 1, slots, per, total, tpb, snap);
        for (uint64_t c = gt_cut(w, slots, per, total, tpb, snap); c < hi;) {
            const uint32_t blk = (uint32_t)(c / tpb);
            const uint64_t t0 = c % tpb, nt = (tpb - t0) < (hi - c) ? (tpb - t0) : (hi - c);
            const uint32_t row0 = (uint32_t)(t0 * kNB);
            const uint64_t rows64 = nt * kNB;
            const uint32_t rows = rows64 < (uint64_t)nb - row0 ? (uint32_t)rows64 : nb - row0;
            if (blk != cur_blk) { close_block(); cur_blk = blk; cur_list = 0; blk_first = n; }
            items[n++] = make_uint4(blk, row0, rows, cur_list++);
            c += nt;
        }
    }
    close_block();
    first[slots] = n;
}

// bias[i] = -0.5 * |b_i|^2 (one 16-lane group per row)
__global__ void rg_gt_bias_kernel(const float *base, uint32_t nb, uint32_t bstride, uint32_t dim, float *bias) {
    const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    if (gid >= nb) return;
    const float *row = base + (size_t)gid * bstride;
    float s = 0.f;
    for (uint32_t j = sub; j < dim; j += 16) s = __builtin_fmaf(row[j], row[j], s);
    for (int o = 8; o; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0) bias[gid] = -0.5f * s;
}

// L2 finalisation: exact squared distance of each of the Kin >= K pairs the approximate ranking kept, re-sort the row by
// (dist, id), write the first K.  Kin > K is the safety margin at the K boundary: the ranking value q.b - |b|^2/2 carries
// fp32 rounding of its own, so a true top-K member may sit a few places below rank K in it (unit-norm embeddings at 10M
// rows have rank-K / rank-K+1 gaps near that rounding).  One wave per query.
template <int ITEMS>
__global__ void __launch_bounds__(64) rg_gt_rescore_kernel(const float *base, uint32_t bstride, const float *queries,
                                                           uint32_t qstride, uint32_t dim, uint32_t nq, uint32_t Kin,
                                                           uint32_t K, uint32_t id_base, const uint32_t *ids_in,
                                                           uint32_t *ids, float *vals) {
    const int lane = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const float *qv = queries + (size_t)q * qstride;
        u64 key[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t e = it * 64 + lane;
            key[it] = ~0ull;
            if (e < Kin) {
                const uint32_t id = ids_in[(size_t)q * Kin + e];
                const float *row = base + (size_t)(id - id_base) * bstride;
                float s = 0.f;
                for (uint32_t j = 0; j < dim; ++j) { const float t = qv[j] - row[j]; s = __builtin_fmaf(t, t, s); }
                key[it] = make_key(s, id, false);
            }
        }
        wave_sort<ITEMS>(key, lane);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t e = it * 64 + lane;
            if (e < K) { ids[(size_t)q * K + e] = (uint32_t)key[it]; vals[(size_t)q * K + e] = key_value(key[it], false); }
        }
    }
}

// The same finalisation with the rows GATHERED COOPERATIVELY (round 6): the kernel above gives every lane a candidate of its own and walks
// that row with scalar loads -- 128 candidates x 2 KB per query at d = 512 read four bytes per lane at a time: 50 ms for 65,536 queries,
// 3.3 % of the whole L2 ground truth (rocprofv3: profiles/r06/gt_d512_l2_65536_trace.txt).  Here a wave takes one query, stages it in LDS,
// and streams the Kin candidate rows four at a time through a ring of R LDS-DMA passes into the scoring routine K1 / K1b use
// (rg_device.h: gather_issue / gather_score) -- 16 lanes per row, coalesced 16-byte loads.  The distance written is therefore exactly
// DistanceL2::compare(row, query, dim) of the reference (distance.h:39-89: its lane order, its folds), the value SearchRoarGraph itself
// would report for that pair.  dim % 8 == 0.
template <int ITEMS, int R>
__global__ void __launch_bounds__(64) rg_gt_rescore_gather_kernel(const float *base, uint32_t bstride, const float *queries, uint32_t qstride, uint32_t dim,
                                                                  uint32_t nq, uint32_t Kin, uint32_t K, uint32_t id_base, const uint32_t *ids_in, uint32_t *ids,
                                                                  float *vals, uint32_t stage_floats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x, g = lane >> 4;
    float *stage = reinterpret_cast<float *>(smem);                 // R passes of stage_floats
    float *qv = stage + (size_t)R * stage_floats;                   // dim
    float *dist = qv + dim;                                         // 64 * ITEMS
    uint32_t *cid = reinterpret_cast<uint32_t *>(dist + 64 * ITEMS);  // 64 * ITEMS
    const uint32_t npass = (Kin + 3u) >> 2, lpp = loads_per_pass(dim);
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        for (uint32_t i = lane; i < dim; i += kWave) qv[i] = queries[(size_t)q * qstride + i];
        for (uint32_t e = lane; e < Kin; e += kWave) cid[e] = ids_in[(size_t)q * Kin + e];
        lds_sync();
        auto issue = [&](uint32_t p, float *buf) {
            const uint32_t c = 4 * p + (uint32_t)g;
            const bool act = c < Kin;
            gather_issue(base + (size_t)(act ? cid[c] - id_base : 0u) * bstride, dim, act, buf, lane);
        };
        for (uint32_t p = 0; p < (uint32_t)R && p < npass; ++p) issue(p, stage + (size_t)p * stage_floats);
        for (uint32_t p = 0; p < npass; ++p) {
            const uint32_t last = min(npass, p + (uint32_t)R) - 1u;
            gather_wait((last - p) * lpp);
            float *buf = stage + (size_t)(p & (R - 1)) * stage_floats;
            const uint32_t c = 4 * p + (uint32_t)g;
            const float d = gather_score<true>(buf, qv, dim, lane);
            if (c < Kin && (lane & 15) == 0) dist[c] = d;
            lds_sync();
            if (p + R < npass) issue(p + R, buf);
        }
        lds_sync();
        u64 key[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t e = it * 64 + lane;
            key[it] = e < Kin ? make_key(dist[e], cid[e], false) : ~0ull;
        }
        wave_sort<ITEMS>(key, lane);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t e = it * 64 + lane;
            if (e < K) { ids[(size_t)q * K + e] = (uint32_t)key[it]; vals[(size_t)q * K + e] = key_value(key[it], false); }
        }
        lds_sync();
    }
}

// K3: merge nlists sorted K-lists per query; one wave per query
template <int ITEMS>
__global__ void __launch_bounds__(64) rg_gt_merge_kernel(const uint32_t *ids_in, const float *vals_in, uint32_t nlists,
                                                         uint32_t nq, uint32_t K, int larger_first, uint32_t *ids,
                                                         float *vals) {
    const int lane = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        u64 key[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t e = it * 64 + lane;
            key[it] = ~0ull;
            if (e < nlists * K) {
                const uint32_t l = e / K, j = e % K;
                const size_t src = ((size_t)l * nq + q) * K + j;
                key[it] = make_key(vals_in[src], ids_in[src], larger_first != 0);
            }
        }
        wave_sort<ITEMS>(key, lane);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t e = it * 64 + lane;
            if (e < K) { ids[(size_t)q * K + e] = (uint32_t)key[it]; vals[(size_t)q * K + e] = key_value(key[it], larger_first != 0); }
        }
    }
}

static size_t gt_lds(uint32_t dim, uint32_t mq, uint32_t bk) {
    return ((size_t)dim * mq + 2 * (size_t)bk * kNB + 2 * mq + 8) * 4;
}

template <int MQ, int ITEMS>
static rg_status launch_gt(const GtParams &P, uint32_t grid, size_t lds, hipStream_t s) {
    auto kern = rg_gt_kernel<MQ, ITEMS>;
    RG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, P);
    RG_HIP(hipGetLastError());
    return RG_OK;
}
template <int MQ>
static rg_status launch_gt_items(int items, const GtParams &P, uint32_t grid, size_t lds, hipStream_t s) {
    switch (items) {
        case 4: return launch_gt<MQ, 4>(P, grid, lds, s);
        case 8: return launch_gt<MQ, 8>(P, grid, lds, s);
        default: return launch_gt<MQ, 16>(P, grid, lds, s);
    }
}
static int items_for(uint32_t n) {  // smallest ITEMS in {4,8,16} with 64*ITEMS >= n
    int it = 4;
    while ((uint32_t)(64 * it) < n && it < 16) it <<= 1;
    return it;
}

}  // namespace rg

using rg::set_error;

namespace rg {

void gt_workspace_free(GtWorkspace *ws) {
    if (!ws) return;
    for (int i = 0; i < 10; ++i) {
        if (ws->p[i]) (void)hipFree(ws->p[i]);
        ws->p[i] = nullptr; ws->cap[i] = 0;
    }
}

// K2 (+ K2b, segment merge) over one shard.  Scratch comes from the stream-ordered allocator, or -- ws != nullptr -- from a
// caller-owned grow-only workspace: the multi-rank driver runs one host thread per rank on its own streams and keeps the
// ranks' scratch apart (several threads allocating and freeing stream-ordered blocks of one device pool while their
// kernels overlap handed a block to a second stream before the first one was done with it on this ROCm).
rg_status gt_shard_ws(const float *d_base, uint32_t nb, uint32_t bstride, const float *d_queries, uint32_t nq,
                      uint32_t qstride, uint32_t dim, int metric, uint32_t K, uint32_t id_base, uint32_t *d_ids,
                      float *d_dists, int device, void *stream, GtWorkspace *ws) {
    hipStream_t s = (hipStream_t)stream;
    if (!d_base || !d_queries || !d_ids || !d_dists) return set_error(RG_ERR_ARG, "null argument");
    if (metric != RG_METRIC_L2 && metric != RG_METRIC_IP && metric != RG_METRIC_COSINE)
        return set_error(RG_ERR_ARG, "Unknown distance type");
    if (dim == 0 || dim % 8 || bstride % 4 || qstride % 4 || bstride < dim || qstride < dim ||
        ((uintptr_t)d_base & 15) || ((uintptr_t)d_queries & 15))
        return set_error(RG_ERR_ARG, "ground truth needs dim % 8 == 0, strides % 4 == 0 and 16-byte aligned buffers");
    if (K == 0 || K > nb) return set_error(RG_ERR_ARG, "K must be in [1, number of base rows in the shard]");
    if (K + kNB > 1024) return set_error(RG_ERR_ARG, "K larger than 896 is not supported");
    if (nq == 0) return RG_OK;
    // L2 ranks by the fp32 value q.b - |b|^2/2 and re-scores the survivors exactly: keep a margin of up to 32 extra
    // survivors (as many as fit the same sort width and the shard) so that rounding of the ranking value at the K
    // boundary cannot drop a true top-K member before the exact re-score (rg_gt_rescore_kernel)
    const uint32_t K_out = K;
    if (metric == RG_METRIC_L2 && !getenv("RG_GT_NO_MARGIN")) {
        uint32_t kin = std::min<uint32_t>({K + 32u, nb, 1024u - kNB});
        while (kin > K && items_for(kin + kNB) != items_for(K + kNB)) --kin;
        K = kin;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(RG_ERR_DEVICE, "no HIP device visible: the gfx950 path cannot run (there is no CPU fallback)");
    RG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RG_HIP(hipGetDeviceProperties(&prop, device));
    const size_t lds_max = 160 * 1024;
    uint32_t bk = 0, mq = 0;
    for (uint32_t cand_mq : {128u, 64u}) {
        for (uint32_t m = 5; m >= 1 && !bk; --m)
            if ((dim / 8) % m == 0 && gt_lds(dim, cand_mq, 8 * m) <= lds_max) { bk = 8 * m; mq = cand_mq; }
        if (bk) break;
    }
    if (!bk) return set_error(RG_ERR_ARG, "dimension too large for the LDS-resident query block");
    const size_t lds = gt_lds(dim, mq, bk);
    const int items = items_for(K + kNB);
    // register-stationary kernel for the BASELINE dimensions when the sort fits ITEMS=4 (K <= 128)
    // (dimension, k-chunk, workgroups per CU) of the register-stationary instantiations: the BASELINE dimensions plus
    // the common embedding widths that satisfy its constraints (A operands dim/2 <= 256 registers, an even number of
    // k-quads per chunk, 2..8 chunks per tile)
    struct RsCfg { uint32_t dim, bk, per_cu; };
    static const RsCfg kRs[] = {{200, 40, 2}, {512, 64, 1}, {96, 48, 2}, {128, 64, 2}, {256, 64, 1}, {384, 64, 1}};
    uint32_t rs_tmw = 0, rs_bk = 0, rs_mqb = 0, rs_per_cu = 1;
    if (items == 4 && !getenv("RG_GT_GENERIC")) {
        for (const RsCfg &c : kRs)
            if (c.dim == dim) { rs_tmw = 1; rs_bk = c.bk; rs_per_cu = c.per_cu; }
        rs_mqb = 128 * rs_tmw;
    }
    // (round 6, measured and removed: 16-row query tiles -- v_mfma_f32_16x16x4_f32, 128 A registers at d = 512, two workgroups of 64 queries
    // per CU; three at d = 200.  SLOWER everywhere: 0.834 against 0.882 of peak at d = 512 IP / 65,536 queries, 0.72 against 0.78 at 10,000,
    // 0.73 against 0.88 at d = 200 -- one B-fragment read per MFMA and a base stream shared by half as many queries cost more than the
    // second workgroup covers: profiles/r06/gt_ab_box2_*.jsonl)
    if (rs_tmw) mq = rs_mqb;
    const uint32_t nblocks = (nq + mq - 1) / mq;
    const uint32_t per_cu = rs_tmw ? rs_per_cu : 1;   // matches the kernel's launch bounds
    const uint32_t slots = (uint32_t)prop.multiProcessorCount * per_cu;
    // Few query blocks (a 10k-query test set is 79 blocks for 512 slots): cut the shard's rows in segments as well, one
    // work item per (block, segment), and merge the per-segment lists with K3.  Segments are whole tiles, keep at least
    // 4096 rows (and K) each, and nseg * K stays within K3's 1024 keys.
    uint32_t nseg = 1, seg_rows = nb;
    if (!getenv("RG_GT_NOSEG") && nblocks * 2 <= slots) {
        nseg = std::min<uint32_t>({slots / nblocks, 1024u / K, 16u, std::max<uint32_t>(1u, nb / std::max<uint32_t>(4096u, K))});
        nseg = std::max<uint32_t>(nseg, 1u);
        seg_rows = ((nb + nseg - 1) / nseg + kNB - 1) / kNB * kNB;
        nseg = (nb + seg_rows - 1) / seg_rows;
        if (nb - (nseg - 1) * seg_rows < K) nseg = 1, seg_rows = nb;   // the last segment must still hold K rows
    }
    uint32_t grid = std::min<uint32_t>(nblocks * nseg, slots);
    // Balanced form (round 4).  The work is a rectangle of nblocks query blocks x tpb base tiles.  Equal work items handed out
    // one by one leave workgroups idle whenever their number is not a multiple of the resident workgroups: 10,000 queries at
    // d = 200 are 79 blocks x 6 segments = 474 items for 512 workgroups, and with the padding of the last block 0.916 of the
    // chip does 0.88-efficient work (0.79 of the MFMA peak measured, against 0.88 at 65,536 queries = 512 blocks).  Instead
    // the rectangle, walked block by block, is cut into `slots` stretches of equal length; a stretch that crosses a block
    // boundary is two items (a cut closer than `snap` tiles to a block boundary moves onto it: every item keeps >= K rows).
    // A block's pieces go to separate lists, merged by K3 -- at most lmax = 1024 / K lists, else the older forms stay.
    uint32_t bal_lists = 0, bal_items = 0;
    uint64_t bal_per = 0, bal_total = 0, bal_tpb = 0, bal_snap = 0;
    if (!getenv("RG_GT_NOBALANCE")) {
        const uint64_t tpb = (nb + kNB - 1) / kNB, total = tpb * nblocks;
        const uint64_t per = (total + slots - 1) / slots;
        const uint64_t snap = std::max<uint64_t>((K + kNB - 1) / kNB, 4);
        // makespan of the older forms: rounds of equal items
        const uint64_t items_old = (uint64_t)nblocks * nseg, tiles_item = (tpb + nseg - 1) / nseg;
        const uint64_t span_old = (items_old + slots - 1) / slots * tiles_item;
        // (measured, scripts/exp/gt_small_batch.py: where the equal items fit ONE round the balanced form is 2 % behind -- an idle
        // workgroup slot leaves its CU's MFMA pipes to the neighbour, and more pieces mean more cold thresholds -- 0.730 vs 0.746 of
        // peak at 10,000 queries; with a partial second round it is 16 % ahead: 0.789 vs 0.681 at 100,000)
        // (round 5: also wherever one round of equal items leaves more than 6 % of the workgroup slots empty -- 391 blocks of 50,000
        // queries on 512 slots ran at 0.67 of peak against 0.83 balanced, 470 items of 30,000 at 0.79 against 0.82, 474 of 10,000 at 0.770
        // against 0.777; a full round -- 8,192 or 16,384 queries -- stays as it is: 0.84 / 0.85 against 0.78 / 0.82,
        // profiles/r05/gt_ab_box11_balance_one.jsonl)
        const bool poor_fill = items_old <= slots && items_old * 100 < (uint64_t)slots * 94;
        if (total >= slots && (getenv("RG_GT_BALANCE_ONE") || poor_fill || (items_old > slots && per + snap < span_old - span_old / 8)) && per > 2 * snap) {
            // the loop of rg_gt_items_kernel, to size and validate the table
            uint32_t n = 0, cur_blk = 0xffffffffu, cur_list = 0, max_list = 0;
            bool ok = true;
            for (uint32_t w = 0; w < slots; ++w) {
                const uint64_t hi = gt_cut(w + 1, slots, per, total, tpb, snap);
                for (uint64_t c = gt_cut(w, slots, per, total, tpb, snap); c < hi;) {
                    const uint32_t blk = (uint32_t)(c / tpb);
                    const uint64_t t0 = c % tpb, nt = std::min<uint64_t>(tpb - t0, hi - c);
                    const uint32_t row0 = (uint32_t)(t0 * kNB), rows = (uint32_t)std::min<uint64_t>(nt * kNB, (uint64_t)nb - row0);
                    if (blk != cur_blk) { cur_blk = blk; cur_list = 0; }
                    max_list = std::max(max_list, ++cur_list);
                    ok = ok && rows >= K;
                    ++n;
                    c += nt;
                }
            }
            if (ok && max_list > 1 && (uint64_t)max_list * K <= 1024) {
                bal_lists = max_list; bal_items = n; bal_per = per; bal_total = total; bal_tpb = tpb; bal_snap = snap;
            }
        }
    }
    if (bal_lists) { nseg = bal_lists; seg_rows = nb; grid = slots; }
    // scratch, released on every exit path (stream-ordered blocks) or kept for the next call (workspace)
    struct Scratch {
        hipStream_t s;
        GtWorkspace *ws;
        void *p[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Scratch() { if (!ws) for (void *q : p) if (q) (void)hipFreeAsync(q, s); }
        hipError_t get(int i, size_t bytes) {
            bytes = std::max<size_t>(bytes, 64);
            if (!ws) return hipMallocAsync(&p[i], bytes, s);
            if (ws->cap[i] < bytes) {
                if (ws->p[i]) { (void)hipStreamSynchronize(s); (void)hipFree(ws->p[i]); ws->p[i] = nullptr; ws->cap[i] = 0; }
                hipError_t e = dev_malloc_retry(&ws->p[i], bytes);
                if (e != hipSuccess) return e;
                ws->cap[i] = bytes;
            }
            p[i] = ws->p[i];
            return hipSuccess;
        }
    } scratch{s, ws};
    float *bias = nullptr;
    u64 *cand = nullptr;
    uint32_t *counter = nullptr;
    float *vals = d_dists;
    uint32_t *ids_k2 = d_ids;            // where K2 (and the segment merge) leave their K-lists
    if (K != K_out) {                    // L2 with a margin: K2 writes K-wide scratch lists, the re-score writes the K_out-wide result
        RG_HIP(scratch.get(5, (size_t)nq * K * 4));
        RG_HIP(scratch.get(6, (size_t)nq * K * 4));
        ids_k2 = static_cast<uint32_t *>(scratch.p[5]);
        vals = static_cast<float *>(scratch.p[6]);
    }
    if (metric == RG_METRIC_L2) {
        RG_HIP(scratch.get(0, (size_t)nb * 4));
        bias = static_cast<float *>(scratch.p[0]);
        hipLaunchKernelGGL(rg_gt_bias_kernel, dim3((nb * 16 + 255) / 256), dim3(256), 0, s, d_base, nb, bstride, dim, bias);
    }
    // candidate buffer of the register-stationary kernel in units of 64 keys: 4 (256 keys: a query sheds down to K after every 29
    // candidates at K = 100), 6 or 8 (d = 200; after 157 / 285) -- RG_GT_CAND
    const int cand_env = getenv("RG_GT_CAND") ? atoi(getenv("RG_GT_CAND")) : 0;
    // default (profiles/r05/gt_ab_box12_buffers_256_384_512.jsonl, d = 200, % of the fp32-MFMA peak with 256 / 384 / 512 keys): 77.0 / 79.0 /
    // 78.1 at 10,000 queries, 80.9 / 81.5 / 80.8 at 30,000, 82.3 / 82.4 / 81.6 at 100,000 -- and 88.6 / 88.2 / 87.3 at 65,536, where a
    // workgroup streams the whole shard for its block and sheds rarely anyway: 384 keys where a block is searched in pieces, 256 otherwise
    // (round 6: 384 keys also wherever 256 would shed after fewer than 16 candidates -- L2's K + 28 = 128 survivors: 0.79 of peak at d = 200 /
    // 65,536 queries against 0.88 for IP, box 3b)
    const int rs_items = (rs_tmw && dim == 200) ? ((cand_env == 8 || cand_env == 6 || cand_env == 4) ? cand_env : (nseg > 1 || bal_lists || K + 16u > 256u - kNB) ? 6 : 4)
                         // d = 512 (round 6): 384 keys.  L2 ranks K + 28 = 128 survivors, which left a 256-key buffer NO room at all -- a
                         // compaction event per candidate: 0.71 -> 0.85 of peak at 65,536 queries, 0.56 -> 0.82 at 30,000, 0.45 -> 0.77 at 10,000; IP
                         // (K = 100) gains too where a block is searched in pieces: 0.78 -> 0.80 at 10,000, 0.827 -> 0.839 at 30,000, level at 65,536
                         // (profiles/r06/gt_ab_box2_d512_{ip,l2}.jsonl).  RG_GT_CAND=4: 256 keys
                         : (rs_tmw && dim == 512 && cand_env != 4) ? 6 : 4;
    RG_HIP(scratch.get(1, (size_t)grid * mq * 64 * (rs_tmw ? rs_items : items) * 8));
    cand = static_cast<u64 *>(scratch.p[1]);
    RG_HIP(scratch.get(2, 128));
    counter = static_cast<uint32_t *>(scratch.p[2]);
    RG_HIP(hipMemsetAsync(counter, 0, 128, s));
    const bool rs_prof = rs_tmw && dim == 200 && getenv("RG_GT_PROF");
    GtParams P;
    P.base = d_base; P.nb = nb; P.bstride = bstride; P.queries = d_queries; P.nq = nq; P.qstride = qstride; P.dim = dim;
    P.bias = bias; P.K = K; P.id_base = id_base; P.out_ids = ids_k2; P.out_vals = vals; P.cand = cand;
    P.counter = counter; P.BK = bk;
    P.diag = getenv("RG_GT_DIAG") ? (uint32_t)atoi(getenv("RG_GT_DIAG")) : 0u;
    P.nseg = nseg; P.seg_rows = seg_rows; P.seg_ids = nullptr; P.seg_vals = nullptr;
    P.items = nullptr; P.item_first = nullptr;
    if (nseg > 1) {
        RG_HIP(scratch.get(3, (size_t)nseg * nq * K * 4));
        RG_HIP(scratch.get(4, (size_t)nseg * nq * K * 4));
        P.seg_ids = static_cast<uint32_t *>(scratch.p[3]);
        P.seg_vals = static_cast<float *>(scratch.p[4]);
    }
    if (bal_lists) {
        // blocks cut into fewer pieces than bal_lists leave lists unwritten: every list starts as K entries that rank last
        RG_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(P.seg_vals), (int)0xff800000u, (size_t)nseg * nq * K, s));   // -inf
        RG_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(P.seg_ids), (int)0xffffffffu, (size_t)nseg * nq * K, s));
        RG_HIP(scratch.get(7, (size_t)bal_items * sizeof(uint4)));
        RG_HIP(scratch.get(8, ((size_t)slots + 1) * 4));
        hipLaunchKernelGGL(rg_gt_items_kernel, dim3(1), dim3(1), 0, s, slots, bal_per, bal_total, bal_tpb, bal_snap, nb, static_cast<uint4 *>(scratch.p[7]),
                           static_cast<uint32_t *>(scratch.p[8]));
        P.items = static_cast<const uint4 *>(scratch.p[7]);
        P.item_first = static_cast<const uint32_t *>(scratch.p[8]);
    }
    P.quota_thr = nullptr;
    if (rs_tmw && nseg > 1 && !getenv("RG_GT_NOSHARE")) {      // quota thresholds between the pieces of a query block (register-stationary kernel)
        const size_t words = (size_t)nblocks * nseg * mq;
        RG_HIP(scratch.get(9, words * 4));
        P.quota_thr = static_cast<uint32_t *>(scratch.p[9]);
        RG_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(P.quota_thr), (int)0xff800000u, words, s));      // -inf: nothing published
    }
    rg_status st;
    if (rs_tmw) {
        const size_t lds_rs = ((size_t)2 * rs_bk * kNB + 2 * rs_mqb + 8 + 256) * 4;
#define RG_RS_LAUNCH_I(D, BKV, WPSV, IT)                                                                                     \
    {                                                                                                                        \
        auto kern = rg_gt_rs_kernel<D, BKV, 1, IT, WPSV>;                                                                    \
        RG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rs)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_rs, s, P);                                                       \
    }
#define RG_RS_LAUNCH(D, BKV, WPSV) RG_RS_LAUNCH_I(D, BKV, WPSV, 4)
        switch (dim) {
            case 200:
                if (rs_prof) {
                    auto kern = rg_gt_rs_kernel<200, 40, 1, 4, 2, true>;
                    RG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rs));
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_rs, s, P);
                } else if (rs_items == 8) RG_RS_LAUNCH_I(200, 40, 2, 8)
                else if (rs_items == 6) RG_RS_LAUNCH_I(200, 40, 2, 6)
                else RG_RS_LAUNCH(200, 40, 2)
                break;
            case 512:
                // (round 6, measured and removed: chunks of 128 floats -- four chunk barriers per tile instead of eight, 128 KB of LDS -- 0.735 against
                // 0.894 of peak at 65,536 queries, 0.65 / 0.79 at 10,000: profiles/r06/gt_ab_box5_d512_bk128_*.jsonl)
                if (rs_items == 6) RG_RS_LAUNCH_I(512, 64, 1, 6)
                else RG_RS_LAUNCH(512, 64, 1)
                break;
            case 96: RG_RS_LAUNCH(96, 48, 2) break;
            case 128: RG_RS_LAUNCH(128, 64, 2) break;
            case 256: RG_RS_LAUNCH(256, 64, 1) break;
            default: RG_RS_LAUNCH(384, 64, 1) break;
        }
#undef RG_RS_LAUNCH
#undef RG_RS_LAUNCH_I
        st = hipGetLastError() == hipSuccess ? RG_OK : set_error(RG_ERR_DEVICE, "K2-RS launch failed");
        if (rs_prof && st == RG_OK) {
            unsigned long long h[8];
            RG_HIP(hipStreamSynchronize(s));
            RG_HIP(hipMemcpy(h, counter + 4, sizeof(h), hipMemcpyDeviceToHost));
            // (wave 0 of every workgroup; clock ticks of 10 ns)
            fprintf(stderr, "[rg_gt prof] grid %u nseg %u items/wg: item %.3f, filter %.3f (rare path %.3f), compactions %.3f [s_memtime ticks x 1e-5: read as shares of item]; per workgroup: tiles %.0f, rare-path tiles %.0f, "
                            "compaction events %.0f, candidate stores (wave 0) %.0f\n", grid, nseg, h[0] * 1e-5 / grid, h[1] * 1e-5 / grid, h[2] * 1e-5 / grid, h[3] * 1e-5 / grid,
                    (double)h[4] / grid, (double)h[5] / grid, (double)h[6] / grid, (double)h[7] / grid);
        }
    } else {
        st = mq == 128 ? launch_gt_items<128>(items, P, grid, lds, s) : launch_gt_items<64>(items, P, grid, lds, s);
    }
    if (st == RG_OK && nseg > 1) {   // per-segment lists (ranking value, larger first) -> the shard's list
        const uint32_t gm = std::min<uint32_t>(nq, 256u * 16u);
        switch (items_for(nseg * K)) {
            case 4: hipLaunchKernelGGL((rg_gt_merge_kernel<4>), dim3(gm), dim3(64), 0, s, P.seg_ids, P.seg_vals, nseg, nq, K, 1, ids_k2, vals); break;
            case 8: hipLaunchKernelGGL((rg_gt_merge_kernel<8>), dim3(gm), dim3(64), 0, s, P.seg_ids, P.seg_vals, nseg, nq, K, 1, ids_k2, vals); break;
            default: hipLaunchKernelGGL((rg_gt_merge_kernel<16>), dim3(gm), dim3(64), 0, s, P.seg_ids, P.seg_vals, nseg, nq, K, 1, ids_k2, vals); break;
        }
    }
    if (st == RG_OK && metric == RG_METRIC_L2) {
        const uint32_t g2 = std::min<uint32_t>(nq, (uint32_t)prop.multiProcessorCount * 16u);
        const int it2 = items_for(K);
        const uint32_t stage_floats = ((dim + 63) / 64) * 256;
        constexpr int RR = 4;
        const size_t lds2 = ((size_t)RR * stage_floats + dim + 2 * 64 * (size_t)it2) * 4;
        if (!getenv("RG_GT_RESCORE_SCALAR") && lds2 <= 64 * 1024) {      // rows gathered cooperatively, the reference's compare() bit for bit
#define RG_RESCORE(IT)                                                                                                                          \
    {                                                                                                                                           \
        auto kern = rg_gt_rescore_gather_kernel<IT, RR>;                                                                                        \
        RG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                 \
        hipLaunchKernelGGL(kern, dim3(g2), dim3(64), lds2, s, d_base, bstride, d_queries, qstride, dim, nq, K, K_out, id_base, ids_k2, d_ids, d_dists, stage_floats); \
    }
            switch (it2) {
                case 4: RG_RESCORE(4) break;
                case 8: RG_RESCORE(8) break;
                default: RG_RESCORE(16) break;
            }
#undef RG_RESCORE
        } else
        switch (it2) {
            case 4: hipLaunchKernelGGL((rg_gt_rescore_kernel<4>), dim3(g2), dim3(64), 0, s, d_base, bstride, d_queries, qstride, dim, nq, K, K_out, id_base, ids_k2, d_ids, d_dists); break;
            case 8: hipLaunchKernelGGL((rg_gt_rescore_kernel<8>), dim3(g2), dim3(64), 0, s, d_base, bstride, d_queries, qstride, dim, nq, K, K_out, id_base, ids_k2, d_ids, d_dists); break;
            default: hipLaunchKernelGGL((rg_gt_rescore_kernel<16>), dim3(g2), dim3(64), 0, s, d_base, bstride, d_queries, qstride, dim, nq, K, K_out, id_base, ids_k2, d_ids, d_dists); break;
        }
    }
    if (st != RG_OK) return st;
    RG_HIP(hipGetLastError());
    return RG_OK;
}

}  // namespace rg

extern "C" {

rg_status rg_gt_shard_dev(const float *d_base, uint32_t nb, uint32_t bstride, const float *d_queries, uint32_t nq,
                          uint32_t qstride, uint32_t dim, int metric, uint32_t K, uint32_t id_base, uint32_t *d_ids,
                          float *d_dists, int device, void *stream) {
    return rg::gt_shard_ws(d_base, nb, bstride, d_queries, nq, qstride, dim, metric, K, id_base, d_ids, d_dists, device, stream, nullptr);
}

rg_status rg_gt_merge_dev(const uint32_t *d_ids_in, const float *d_dists_in, uint32_t nlists, uint32_t nq, uint32_t K,
                          int metric, uint32_t *d_ids, float *d_dists, int device, void *stream) {
    using namespace rg;
    hipStream_t s = (hipStream_t)stream;
    if (!d_ids_in || !d_dists_in || !d_ids || !d_dists) return set_error(RG_ERR_ARG, "null argument");
    if (nlists == 0 || K == 0) return set_error(RG_ERR_ARG, "nlists and K must be positive");
    if ((uint64_t)nlists * K > 1024) return set_error(RG_ERR_ARG, "nlists * K larger than 1024 is not supported");
    if (nq == 0) return RG_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(RG_ERR_DEVICE, "no HIP device visible: the gfx950 path cannot run (there is no CPU fallback)");
    RG_HIP(hipSetDevice(device));
    const int lf = metric == RG_METRIC_L2 ? 0 : 1;
    const uint32_t grid = std::min<uint32_t>(nq, 256u * 16u);
    switch (items_for(nlists * K)) {
        case 4: hipLaunchKernelGGL((rg_gt_merge_kernel<4>), dim3(grid), dim3(64), 0, s, d_ids_in, d_dists_in, nlists, nq, K, lf, d_ids, d_dists); break;
        case 8: hipLaunchKernelGGL((rg_gt_merge_kernel<8>), dim3(grid), dim3(64), 0, s, d_ids_in, d_dists_in, nlists, nq, K, lf, d_ids, d_dists); break;
        default: hipLaunchKernelGGL((rg_gt_merge_kernel<16>), dim3(grid), dim3(64), 0, s, d_ids_in, d_dists_in, nlists, nq, K, lf, d_ids, d_dists); break;
    }
    RG_HIP(hipGetLastError());
    return RG_OK;
}

/* rg_groundtruth_mem / rg_groundtruth (whole job, several GPUs of this process) live in rg_gt_dist.hip */

}  // extern "C"
