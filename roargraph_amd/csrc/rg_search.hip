// rg_search.hip -- host side of the search path and its small kernels.
//
//   K1  rg_search_kernel   persistent beam search (rg_search_kernel.h, instantiated in rg_search_inst_*.hip)
//   K4  rg_distinct_kernel exact distinct count of a query's id log (the reference's cmps)
//   K1b rg_score_kernel    batched Distance::compare (include/efanna2e/distance.h:18)
//
// Host state: the rg_index is immutable after open; every launch-time buffer lives in a per-stream SearchCtx handed out
// under the index mutex (rg_index_struct.h), so searches on distinct streams / from distinct host threads do not share
// anything mutable, and several batches may be in flight on one stream before rg_search_wait collects them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rg.h"
#include "rg_device.h"
#include "rg_internal.h"
#include "rg_search_kernel.h"
#include "rg_index_struct.h"
#include "rg_mem.h"

namespace rg {

// RG_TRACE_STALE=1: name a HIP error that an earlier call left behind (hipGetLastError reads AND clears it)
static void trace_stale(const char *where) {
    static const bool on = getenv("RG_TRACE_STALE") != nullptr;
    if (!on) return;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) fprintf(stderr, "[rg] stale HIP error at %s: %s\n", where, hipGetErrorString(e));
}

// the library's plain allocations: a refused request is tried once more after the allocator's cache went back to the device (rg_mem.hip)
template <typename T>
static inline hipError_t rg_malloc(T **p, size_t bytes) { return dev_malloc_retry(reinterpret_cast<void **>(p), bytes); }

// device allocation released on every exit path of a host wrapper
template <typename T>
struct DevBuf {
    T *p = nullptr;
    hipError_t alloc(size_t n) { return dev_malloc_retry(reinterpret_cast<void **>(&p), std::max<size_t>(n * sizeof(T), 16)); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    T *release() { T *r = p; p = nullptr; return r; }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

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
    // Queries are dealt round robin (logs of one launch are about equally long) and the two totals leave the workgroup
    // once, at the end: a shared work counter plus two totals per query were 30,000 same-address atomics per launch, which
    // the L2 serialises -- 0.27 ms of a 3.8 ms step at L_pq = 50 whatever the workgroup shape.
    unsigned long long tot_n = 0, tot_d = 0;
    (void)work; (void)s_q;
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const uint32_t n = qlog_n[q];
        if (n & 0x80000000u) continue;       // counted by the wave that ran the query (K1, narrow beams): out_cmps[q] is final
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
                tot_n += n;          // evaluations performed / distinct nodes: how much the forgetful filter re-scored
                tot_d += s_cnt;      // (search_wait looks at it)
            }
        }
        __syncthreads();
    }
    if (tid == 0 && tot_n) {
        atomicAdd(&totals[0], tot_n);
        atomicAdd(&totals[1], tot_d);
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

// rg_search_reuse_stats: mark every row id of the id logs in a bitmap / count the marks
__global__ void rg_log_mark_kernel(const uint32_t *__restrict__ qlog, uint32_t logcap, const uint32_t *__restrict__ qlog_n, uint32_t nq,
                                   uint32_t *__restrict__ bitmap, unsigned long long *__restrict__ total, uint32_t *__restrict__ row_counts) {
    unsigned long long mine = 0;
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const uint32_t n = min(qlog_n[q] & 0x7fffffffu, logcap);      // (top bit: counted inside K1)
        const uint32_t *log = qlog + (size_t)q * logcap;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint32_t id = log[i];
            atomicOr(&bitmap[id >> 5], 1u << (id & 31u));
            if (row_counts) atomicAdd(&row_counts[id], 1u);
        }
        if (threadIdx.x == 0) mine += n;
    }
    if (threadIdx.x == 0 && mine) atomicAdd(total, mine);
}
__global__ void rg_bitmap_count_kernel(const uint32_t *__restrict__ bitmap, size_t words, unsigned long long *__restrict__ total) {
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) mine += __popc(bitmap[i]);
    for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(total, mine);
}

// Shared frontier (third mode; knob "shared_frontier"): out[q][c] = compare(base[ids[c]], queries[q]) for the S
// rows every query of a batch scores first -- the entry point and its neighbours -- with the exact routine K1 and K1b use
// (same bits).  One wave per query at a time, the query staged in LDS once, the S rows (hot in the L2s: every wave reads
// the same 57 KB) streamed through a ring of R passes of four.
template <bool L2, int R>
__global__ void __launch_bounds__(64) rg_front_score_kernel(const float *__restrict__ base, uint32_t stride, uint32_t dim,
                                                            const float *__restrict__ queries, uint32_t nq, uint32_t qstride,
                                                            const uint32_t *__restrict__ ids, uint32_t n, float *__restrict__ out,
                                                            uint32_t ostride, uint32_t stage_floats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x, g = lane >> 4;
    float *stage = reinterpret_cast<float *>(smem);
    float *qv = stage + (size_t)R * stage_floats;
    const uint32_t npass = (n + 3u) >> 2, lpp = loads_per_pass(dim);
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        for (uint32_t i = lane; i < dim; i += kWave) qv[i] = queries[(size_t)q * qstride + i];
        wave_sync();
        auto issue = [&](uint32_t p, float *buf) {
            const uint32_t c = 4 * p + g;
            const bool act = c < n;
            gather_issue(base + (size_t)(act ? ids[c] : 0u) * stride, dim, act, buf, lane);
        };
        for (uint32_t p = 0; p < (uint32_t)R && p < npass; ++p) issue(p, stage + (size_t)p * stage_floats);
        for (uint32_t p = 0; p < npass; ++p) {
            const uint32_t last = min(npass, p + (uint32_t)R) - 1u;
            gather_wait((last - p) * lpp);
            float *buf = stage + (size_t)(p & (R - 1)) * stage_floats;
            const uint32_t c = 4 * p + g;
            const float d = gather_score<L2>(buf, qv, dim, lane);
            if (c < n && (lane & 15) == 0) out[(size_t)q * ostride + c] = d;
            lds_sync();
            if (p + R < npass) issue(p + R, buf);
        }
        wave_sync();
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

// CSR -> ELL ([deg, ids...] per node at a fixed stride), one wave per node.  *dups is set when a list -- of any length --
// names a node twice (the file format does not forbid it, index_bipartite.cpp:2097-2117; the reference's own builds never
// produce it): the look-ahead form of K1 tests a hop's neighbours against the visited tags in one step and would score
// such a node twice, so those indexes keep the returning-atomic form, which orders the two tests.
__global__ void rg_csr_to_ell_kernel(const uint64_t *offsets, const uint32_t *nbrs, uint32_t nd, uint32_t *ell,
                                     uint32_t ell_stride, uint32_t *dups) {
    const int lane = threadIdx.x & 63;
    const uint32_t wpb = blockDim.x / 64;
    for (uint32_t node = blockIdx.x * wpb + threadIdx.x / 64; node < nd; node += gridDim.x * wpb) {
        const uint64_t o0 = offsets[node];
        const uint32_t deg = (uint32_t)(offsets[node + 1] - o0);
        uint32_t *row = ell + (size_t)node * ell_stride;
        if (lane == 0) row[0] = deg;
        for (uint32_t j = lane; j < ell_stride - 1; j += 64) row[1 + j] = j < deg ? nbrs[o0 + j] : 0u;
        if (deg > 1u) {
            // every degree (rows of 64 .. 126 neighbours take the look-ahead form's two-read step, longer ones its general
            // path with plain byte tags: both test a whole chunk in one step): entry i against every entry before it, the
            // row 64 entries at a time
            bool twice = false;
            for (uint32_t c0 = 0; c0 < deg; c0 += 64u) {
                const uint32_t i = c0 + (uint32_t)lane;
                const uint32_t mine = i < deg ? nbrs[o0 + i] : 0xffffffffu;
                for (uint32_t b0 = 0; b0 <= c0; b0 += 64u) {
                    const uint32_t other = b0 + (uint32_t)lane < deg ? nbrs[o0 + b0 + lane] : 0xffffffffu;
                    const uint32_t nj = min(64u, deg - b0);
                    for (uint32_t j = 0; j < nj; ++j) twice = twice || (i < deg && i > b0 + j && readlane_u(other, (int)j) == mine);
                }
            }
            if (__ballot(twice) != 0ull && lane == 0) atomicOr(dups, 1u);
        }
    }
}

// in-degree of every node / min(255, in-degree of the neighbour) into the top byte of every ELL neighbour word (indexes of
// at most 2^24 nodes; SearchParams::id_mask)
__global__ void rg_indeg_kernel(const uint32_t *__restrict__ nbrs, uint64_t ne, uint32_t *__restrict__ indeg) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (size_t)gridDim.x * blockDim.x) atomicAdd(&indeg[nbrs[e]], 1u);
}
// top byte of a neighbour word: low nibble = min(15, in-degree of the neighbour) (admission rule of the LDS filter), high nibble =
// its hub level (rg_search_kernel.h, SearchParams::hub_m; 15 = never a hub)
__global__ void rg_ell_tag_kernel(uint32_t *__restrict__ ell, uint32_t nd, uint32_t ell_stride, const uint32_t *__restrict__ indeg,
                                  const uint8_t *__restrict__ hub_lvl) {
    const size_t total = (size_t)nd * ell_stride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / ell_stride;
        const uint32_t j = (uint32_t)(i - row * ell_stride);
        if (j == 0 || j > ell[row * ell_stride]) continue;      // word 0 = degree; words beyond it are padding
        const uint32_t id = ell[i];
        ell[i] = id | (min(15u, indeg[id]) << 24) | ((hub_lvl ? (uint32_t)hub_lvl[id] : 15u) << 28);
    }
}

// ---- hub levels (round 5).  A visited bitmap of 2^m bits in LDS gives bit p to ONE node: the node of highest in-degree among those
// whose hashed id falls on p = (id * 0x9E3779B1) >> (32 - m), the smaller id on a tie -- the node met most often per bit spent.
// Positions nest over m, so the owner of a position at 2^m bits owns its position at every larger size; level = the smallest m
// (kHubMinM ... kHubMaxM) at which the node owns a bit.  Key of a node: (in-degree, ~id), the largest key wins a position.
constexpr uint32_t kHubMinM = 8, kHubMaxM = 22, kHubMult = 0x9E3779B1u;
__global__ void rg_hub_key_kernel(const uint32_t *__restrict__ indeg, uint32_t nd, unsigned long long *__restrict__ finest) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nd; v += (size_t)gridDim.x * blockDim.x) {
        const uint32_t dg = indeg[v];
        if (dg == 0) continue;                                  // never met as a neighbour
        atomicMax(&finest[((uint32_t)v * kHubMult) >> (32u - kHubMaxM)], ((unsigned long long)dg << 32) | (0xffffffffu - (uint32_t)v));
    }
}
// tables of all sizes in one array: level m at words [2^m, 2^(m+1)); level m - 1 from level m
__global__ void rg_hub_reduce_kernel(unsigned long long *__restrict__ tabs, uint32_t m) {
    const uint32_t n = 1u << (m - 1u);
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const unsigned long long a = tabs[(2u << (m - 1u)) + 2u * p], b = tabs[(2u << (m - 1u)) + 2u * p + 1u];
        tabs[n + p] = a > b ? a : b;
    }
}
__global__ void rg_hub_level_kernel(const uint32_t *__restrict__ indeg, uint32_t nd, const unsigned long long *__restrict__ tabs, uint8_t *__restrict__ lvl) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nd; v += (size_t)gridDim.x * blockDim.x) {
        const uint32_t dg = indeg[v];
        uint32_t l = 15u;
        if (dg) {
            const unsigned long long key = ((unsigned long long)dg << 32) | (0xffffffffu - (uint32_t)v);
            const uint32_t x = (uint32_t)v * kHubMult;
            for (uint32_t m = kHubMinM; m <= kHubMaxM; ++m)
                if (tabs[(1u << m) + (x >> (32u - m))] == key) { l = m - kHubMinM; break; }
        }
        lvl[v] = (uint8_t)l;
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


// ---- CalculateProjectionep (src/index_bipartite.cpp:2004-2041): entry point = the base row nearest (squared L2) to the
// centroid.  The reference's centroid is a float sum over the rows IN INDEX ORDER per dimension (:2008-2012) and that order
// fixes the bits, so the sum stays a serial chain per dimension; what the GPU adds is width: every workgroup owns 32
// dimensions, all of its 256 threads stream the rows' 128-byte slices into a double-buffered LDS tile, and 32 threads walk
// the tile row by row.  The distance pass is one thread per row with the reference's j-order sum, and the arg-min keeps
// the first of equal distances (:2031-2035) through a 64-bit (distance bits, index) atomicMin.
__global__ void __launch_bounds__(256) rg_centroid_sum_kernel(const float *__restrict__ base, uint32_t nd, uint32_t dim, uint32_t stride,
                                                              float *__restrict__ sum) {
    __shared__ float tile[2][256][33];
    const int t = threadIdx.x;
    const uint32_t d0 = blockIdx.x * 32u, nt = (nd + 255u) / 256u;
    float4 r[8];
    auto load = [&](uint32_t tl) {
        const uint32_t row = tl * 256u + (uint32_t)t;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t c = d0 + 4u * (uint32_t)k;
            r[k] = (row < nd && c < dim) ? *reinterpret_cast<const float4 *>(base + (size_t)row * stride + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            tile[buf][t][4 * k + 0] = r[k].x; tile[buf][t][4 * k + 1] = r[k].y;
            tile[buf][t][4 * k + 2] = r[k].z; tile[buf][t][4 * k + 3] = r[k].w;
        }
    };
    load(0);
    store(0);
    __syncthreads();
    float acc = 0.0f;
    for (uint32_t tl = 0; tl < nt; ++tl) {
        if (tl + 1 < nt) load(tl + 1);                       // next tile's rows fly while this one is summed
        if (t < 32) {
            const uint32_t rows = min(256u, nd - tl * 256u);
            for (uint32_t i = 0; i < rows; ++i) acc += tile[tl & 1u][i][t];
        }
        __syncthreads();
        if (tl + 1 < nt) store((int)((tl + 1) & 1u));
        __syncthreads();
    }
    if (t < 32 && d0 + (uint32_t)t < dim) sum[d0 + t] = acc;
}

__global__ void __launch_bounds__(256) rg_centroid_argmin_kernel(const float *__restrict__ base, uint32_t nd, uint32_t dim, uint32_t stride,
                                                                 const float *__restrict__ center, unsigned long long *best) {
    extern __shared__ float cen[];
    for (uint32_t j = threadIdx.x; j < dim; j += blockDim.x) cen[j] = center[j];
    __syncthreads();
    unsigned long long mine = ~0ull;
    for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < nd; row += gridDim.x * blockDim.x) {
        const float *x = base + (size_t)row * stride;
        float diff = 0.0f;
        for (uint32_t j = 0; j < dim; j += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + j);
            float t0 = cen[j] - v.x; diff += t0 * t0;
            t0 = cen[j + 1] - v.y; diff += t0 * t0;
            t0 = cen[j + 2] - v.z; diff += t0 * t0;
            t0 = cen[j + 3] - v.w; diff += t0 * t0;
        }
        const unsigned long long key = ((unsigned long long)__float_as_uint(diff) << 32) | row;   // diff >= 0: bits order like values
        mine = min(mine, key);
    }
    for (int o = 32; o; o >>= 1) {
        const unsigned long long other = ((unsigned long long)(uint32_t)__shfl_xor((int)(mine >> 32), o, 64) << 32) |
                                         (uint32_t)__shfl_xor((int)(mine & 0xffffffffu), o, 64);
        mine = min(mine, other);
    }
    if ((threadIdx.x & 63) == 0) atomicMin(best, mine);
}

// split rows: main part of every row at a whole-line stride / per-edge tails in adjacency order (slot ne = entry point) /
// first edge of every node
__global__ void __launch_bounds__(256) rg_split_main_kernel(const float *__restrict__ base, uint32_t nd, uint32_t stride, uint32_t main_dim,
                                                            float *__restrict__ out) {
    const uint32_t per = main_dim / 4u;
    const size_t total = (size_t)nd * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / per;
        const uint32_t c = (uint32_t)(i - row * per);
        reinterpret_cast<float4 *>(out)[i] = *reinterpret_cast<const float4 *>(base + row * stride + 4u * c);
    }
}
__global__ void __launch_bounds__(256) rg_split_tail_kernel(const float *__restrict__ base, uint32_t stride, uint32_t main_dim, uint32_t tail_dim,
                                                            const uint32_t *__restrict__ nbrs, uint64_t ne, uint32_t ep, float *__restrict__ etail) {
    const uint32_t per = tail_dim / 4u;
    const size_t total = (size_t)(ne + 1) * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / per;
        const uint32_t c = (uint32_t)(i - e * per);
        const uint32_t row = e < ne ? nbrs[e] : ep;
        reinterpret_cast<float4 *>(etail)[i] = *reinterpret_cast<const float4 *>(base + (size_t)row * stride + main_dim + 4u * c);
    }
}
__global__ void __launch_bounds__(256) rg_tail_off_kernel(const uint64_t *__restrict__ off, uint32_t nd, uint32_t *__restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += gridDim.x * blockDim.x) out[i] = (uint32_t)off[i];
}

// -------------------------------------------------------------------------------------------------- host
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
        {   // (large buffers: balanced over the memory classes of the device, rg_mem.hip)
            rg_status as = dev_alloc_t(ix->device, (size_t)ix->nd * es, &ix->d_ell);
            if (as != RG_OK) return as;
            ix->n_plain_allocs += dev_last_plain() ? 1 : 0;
        }
        RG_HIP(hipMemset(d_stat, 0, 4));
        hipLaunchKernelGGL(rg_csr_to_ell_kernel, dim3(4096), dim3(256), 0, 0, d_off, d_nb, ix->nd, ix->d_ell, es, d_stat);
        RG_HIP(hipDeviceSynchronize());
        uint32_t dups = 0;
        RG_HIP(hipMemcpy(&dups, d_stat, 4, hipMemcpyDeviceToHost));
        ix->adj_dups = dups != 0;
        if (ix->nd <= (1u << 24) && ne > 0) {   // in-degree tags for the admission rule of the LDS visited filter
            DevBuf<uint32_t> indeg;
            if (indeg.alloc(ix->nd) == hipSuccess && hipMemset(indeg.p, 0, (size_t)ix->nd * 4) == hipSuccess) {
                {   // (not this launch's: the runtime keeps the last error of earlier, unrelated calls; RG_TRACE_STALE=1 names it)
                    const hipError_t stale = hipGetLastError();
                    if (stale != hipSuccess && getenv("RG_TRACE_STALE")) fprintf(stderr, "[rg_index_open] stale HIP error before tagging: %s\n", hipGetErrorString(stale));
                }
                hipLaunchKernelGGL(rg_indeg_kernel, dim3(4096), dim3(256), 0, 0, d_nb, (uint64_t)ne, indeg.p);
                // hub levels: 64 MB of position tables for a moment (optional: without them no node is a hub, RG_HUB_BITS=0: never)
                DevBuf<unsigned long long> tabs;
                DevBuf<uint8_t> lvl;
                const char *henv = getenv("RG_HUB_BITS");
                bool hubs = !(henv && atoi(henv) == 0) && tabs.alloc((size_t)2 << kHubMaxM) == hipSuccess && lvl.alloc(ix->nd) == hipSuccess &&
                            hipMemset(tabs.p, 0, ((size_t)2 << kHubMaxM) * 8) == hipSuccess;
                if (hubs) {
                    hipLaunchKernelGGL(rg_hub_key_kernel, dim3(4096), dim3(256), 0, 0, indeg.p, ix->nd, tabs.p + ((size_t)1 << kHubMaxM));
                    for (uint32_t m = kHubMaxM; m > kHubMinM; --m)
                        hipLaunchKernelGGL(rg_hub_reduce_kernel, dim3(std::max(1u, std::min(4096u, (1u << (m - 1u)) / 256u))), dim3(256), 0, 0, tabs.p, m);
                    hipLaunchKernelGGL(rg_hub_level_kernel, dim3(4096), dim3(256), 0, 0, indeg.p, ix->nd, tabs.p, lvl.p);
                } else (void)hipGetLastError();
                hipLaunchKernelGGL(rg_ell_tag_kernel, dim3(8192), dim3(256), 0, 0, ix->d_ell, ix->nd, es, indeg.p, hubs ? lvl.p : nullptr);
                const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
                if (e1 == hipSuccess && e2 == hipSuccess) { ix->ell_tagged = true; ix->hub_levels = hubs; }
                else return set_error(RG_ERR_DEVICE, std::string("tagging the adjacency rows failed: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            } else (void)hipGetLastError();
        }
        // split rows: an 800-B row (d = 200) spans seven 128-B lines wherever it starts; its first 192 elements at a 768-B
        // stride span six, and the 8-element tails, stored per edge in adjacency order, are read from ceil(deg/4) lines
        // per hop instead of one more line per fresh neighbour.  7.7 GB + 32 B per edge at 10M rows.
        const char *env = getenv("RG_SPLIT_ROWS");
        if (ix->dim == 200 && ne + 1 < 0xffffffffull && !(env && atoi(env) == 0)) {
            ix->main_dim = 192; ix->tail_dim = 8;
            // optional: an index that cannot afford the copy (or whose kernels fail) searches the base itself
            bool ok = dev_alloc_t(ix->device, (size_t)ix->nd * ix->main_dim, &ix->d_main) == RG_OK;
            ix->n_plain_allocs += ok && dev_last_plain() ? 1 : 0;
            ok = ok && dev_alloc_t(ix->device, (size_t)(ne + 1) * ix->tail_dim, &ix->d_etail) == RG_OK;
            ix->n_plain_allocs += ok && dev_last_plain() ? 1 : 0;
            ok = ok && rg_malloc(&ix->d_tail_off, (size_t)ix->nd * 4) == hipSuccess;
            if (ok) {
                hipLaunchKernelGGL(rg_split_main_kernel, dim3(8192), dim3(256), 0, 0, ix->d_base, ix->nd, ix->stride, ix->main_dim, ix->d_main);
                hipLaunchKernelGGL(rg_split_tail_kernel, dim3(8192), dim3(256), 0, 0, ix->d_base, ix->stride, ix->main_dim, ix->tail_dim, d_nb,
                                   (uint64_t)ne, ix->ep, ix->d_etail);
                hipLaunchKernelGGL(rg_tail_off_kernel, dim3(2048), dim3(256), 0, 0, d_off, ix->nd, ix->d_tail_off);
                ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
            }
            if (!ok) {
                (void)hipGetLastError();
                dev_free(ix->d_main);
                dev_free(ix->d_etail);
                if (ix->d_tail_off) (void)hipFree(ix->d_tail_off);
                ix->d_main = ix->d_etail = nullptr;
                ix->d_tail_off = nullptr;
            }
        }
    } else {
        RG_HIP(rg_malloc(&ix->d_offsets, ((size_t)ix->nd + 1) * 8));
        RG_HIP(rg_malloc(&ix->d_nbrs, std::max<size_t>(ne * 4, 4)));
        RG_HIP(hipMemcpy(ix->d_offsets, d_off, ((size_t)ix->nd + 1) * 8, hipMemcpyDeviceToDevice));
        RG_HIP(hipMemcpy(ix->d_nbrs, d_nb, ne * 4, hipMemcpyDeviceToDevice));
    }
    // The rows K1 gathers: the split copy made above (d = 200) is balanced over the memory classes; a base the library
    // loaded itself is too (rg_index_open_mem).  A CALLER's device base is one plain allocation -- one class -- so, where
    // the searches will read it directly and the device has the room, the index keeps a balanced copy of its own
    // (RG_COPY_BASE=0: never; the caller's buffer is then only read at open).
    {
        const size_t bytes = (size_t)ix->nd * ix->stride * 4;
        const char *env = getenv("RG_COPY_BASE");
        size_t free_b = 0, total_b = 0;
        if (!ix->own_base && !ix->d_main && bytes >= ((size_t)2 << 30) && !(env && atoi(env) == 0) &&
            hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 2 * bytes + ((size_t)8 << 30)) {
            float *copy = nullptr;
            if (dev_alloc_t(ix->device, (size_t)ix->nd * ix->stride, &copy) == RG_OK) {
                ix->n_plain_allocs += dev_last_plain() ? 1 : 0;
                if (hipMemcpy(copy, ix->d_base, bytes, hipMemcpyDeviceToDevice) == hipSuccess) { ix->d_base = copy; ix->own_base = true; ix->base_copied = true; }
                else { (void)hipGetLastError(); dev_free(copy); }
            }
        }
    }
    {   // the shared frontier: the entry point and its neighbours, in adjacency order
        uint64_t o[2] = {0, 0};
        RG_HIP(hipMemcpy(o, d_off + ix->ep, 16, hipMemcpyDeviceToHost));
        const uint32_t dg = (uint32_t)(o[1] - o[0]);
        ix->front_n = 1 + dg;
        RG_HIP(rg_malloc(&ix->d_front_ids, (size_t)ix->front_n * 4));
        RG_HIP(hipMemcpy(ix->d_front_ids, &ix->ep, 4, hipMemcpyHostToDevice));
        if (dg) RG_HIP(hipMemcpy(ix->d_front_ids + 1, d_nb + o[0], (size_t)dg * 4, hipMemcpyDeviceToDevice));
    }
    hipDeviceProp_t prop;
    RG_HIP(hipGetDeviceProperties(&prop, ix->device));
    ix->num_cu = prop.multiProcessorCount;
    return RG_OK;
}

// ---- per-stream contexts ---------------------------------------------------------------------------------------
static void free_batch(Batch *b) {
    if (!b) return;
    if (b->d_stat) (void)hipFree(b->d_stat);
    if (b->h_stat) (void)hipHostFree(b->h_stat);
    if (b->d_ovf) (void)hipFree(b->d_ovf);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    delete b;
}

static void free_ctx(SearchCtx *cx) {
    if (!cx) return;
    for (Batch *b : cx->pending) free_batch(b);
    for (Batch *b : cx->spare) free_batch(b);
    if (cx->own) (void)hipStreamDestroy(cx->own);
    if (cx->h_pin) (void)hipHostFree(cx->h_pin);
    if (cx->d_front) (void)hipFree(cx->d_front);
    void *bufs[] = {cx->d_counter, cx->d_scratch_stat, cx->d_epoch, cx->d_epoch8, cx->d_qlog_n,
                    cx->d_q, cx->d_dist, cx->d_ids, cx->d_ch};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    dev_free(cx->d_visited);
    dev_free(cx->d_vtags);
    dev_free(cx->d_qlog);
    delete cx;
}

static rg_status new_ctx(SearchCtx **out) {
    SearchCtx *cx = new SearchCtx();
    hipError_t e = rg_malloc(&cx->d_counter, 64);
    if (e == hipSuccess) e = rg_malloc(&cx->d_scratch_stat, 64);
    if (e != hipSuccess) { free_ctx(cx); return set_error(RG_ERR_DEVICE, hipGetErrorString(e)); }
    *out = cx;
    return RG_OK;
}

// the context serving `s`: the one already keyed to it, else an idle one, else a new one.  `priv`: a host-form call
// wants a context nobody else touches and a stream of its own (concurrent rg_search calls from several host threads).
static rg_status acquire_ctx(rg_index *ix, hipStream_t s, bool priv, SearchCtx **out) {
    std::lock_guard<std::mutex> lk(ix->mu);
    SearchCtx *cx = nullptr;
    if (!priv)
        for (SearchCtx *c : ix->ctxs)
            if (c->keyed && !c->reserved && c->key == s) { cx = c; break; }
    if (!cx)
        for (SearchCtx *c : ix->ctxs)
            if (!c->keyed && !c->reserved) { cx = c; break; }
    if (!cx) {
        rg_status st = new_ctx(&cx);
        if (st != RG_OK) return st;
        ix->ctxs.push_back(cx);
    }
    if (priv) {
        if (!cx->own) {
            if (hipStreamCreateWithFlags(&cx->own, hipStreamNonBlocking) != hipSuccess) return set_error(RG_ERR_DEVICE, "cannot create a stream");
        }
        cx->reserved = true;
        cx->key = cx->own;
    } else {
        cx->key = s;
    }
    cx->keyed = true;
    *out = cx;
    return RG_OK;
}

static void release_ctx(rg_index *ix, SearchCtx *cx) {
    std::lock_guard<std::mutex> lk(ix->mu);
    if (cx->pending.empty()) { cx->keyed = false; cx->reserved = false; }
}

static rg_status take_batch(SearchCtx *cx, uint32_t nq, Batch **out) {
    Batch *b = nullptr;
    if (!cx->spare.empty()) { b = cx->spare.back(); cx->spare.pop_back(); }
    else b = new Batch();
    hipError_t e = hipSuccess;
    if (!b->d_stat) e = rg_malloc(&b->d_stat, 64);
    if (e == hipSuccess && !b->h_stat) e = hipHostMalloc(&b->h_stat, 64);
    if (e == hipSuccess && b->ovf_cap < nq + 2) {
        if (b->d_ovf) (void)hipFree(b->d_ovf);
        b->d_ovf = nullptr; b->ovf_cap = 0;
        e = rg_malloc(&b->d_ovf, ((size_t)nq + 2) * 4);
        if (e == hipSuccess) b->ovf_cap = nq + 2;
    }
    if (e != hipSuccess) { free_batch(b); return set_error(RG_ERR_DEVICE, hipGetErrorString(e)); }
    b->counted = b->timed = b->is_trial = b->cold = false;
    b->mode = 2;
    *out = b;
    return RG_OK;
}

// ---- launch planning ----------------------------------------------------------------------------------------------
static uint32_t id_bits_of(uint32_t nd) {
    uint32_t b = 1;
    while (b < 32 && (1ull << b) < nd) ++b;
    return b;
}
static uint32_t filter_log2_of(const rg_index *ix, int automatic) {  // remainder must fit 15 bits
    const uint32_t bits = id_bits_of(ix->nd);
    uint32_t t = (uint32_t)std::max(4, std::min(14, ix->filter_log2 > 0 ? ix->filter_log2 : automatic));
    if (bits > t + 15) t = bits - 15;
    return std::min(t, bits);
}
// dimensions with a register-query instantiation of K1 (the BASELINE configs); ELL adjacency only
static int dimc_of(const rg_index *ix) {
    return (ix->d_ell != nullptr && (ix->dim == 200 || ix->dim == 512) && !ix->query_in_lds) ? (int)ix->dim : 0;
}
static size_t stage_pass_floats(const rg_index *ix, bool bf) {
    return bf ? (size_t)((ix->dim + 127) / 128) * 256 : (size_t)((ix->dim + 63) / 64) * 256;
}
// gather form of a launch (rg_search_kernel.h, GF): compute-layout loads wherever an instantiation exists -- d = 200 with
// four or eight register sets, d = 512 with two -- unless the knob "gather_form" says 0
static int gather_form_of(const rg_index *ix, int R, bool bf) {
    const int dc = (ix->d_ell != nullptr && (ix->dim == 200 || ix->dim == 512) && !ix->query_in_lds) ? (int)ix->dim : 0;
    if (ix->gather_form == 0 || bf) return 0;
    return ((dc == 200 && (R == 4 || R == 8)) || (dc == 512 && R == 2)) ? 1 : 0;
}
static size_t stage_total_floats(const rg_index *ix, int R, bool bf) {   // the fast mode's exact re-rank needs one fp32 pass
    // register-staged instantiations (compile-time dimension, fp32): rows in flight live in VGPRs; the bounce form needs one
    // 1-KiB LDS block at a time (bounce_score_q), the compute-layout form none
    if (dimc_of(ix) && !bf) return gather_form_of(ix, R, bf) ? 0 : 256;
    return std::max((size_t)R * stage_pass_floats(ix, bf), (size_t)((ix->dim + 63) / 64) * 256);
}
static size_t search_lds_bytes(const rg_index *ix, uint32_t L, int R, int mode, bool bf, int filter_auto) {
    size_t b = stage_total_floats(ix, R, bf) * 4 + (dimc_of(ix) ? 0 : (size_t)ix->dim * 4) + kCand * 4 + kCand * 4 + 2 * kWave * 4 + (size_t)L * 8;
    // (the 128-word id-log line exists in the forms that may log: the LDS filter and the exact LDS set, not the exact words)
    if (mode != 0 || ix->exact_filter) b += (mode != 0 ? 128 * 4 : 0) + std::max<size_t>(16, (size_t)2 << filter_log2_of(ix, filter_auto));
    return (b + 15) / 16 * 16;
}
// bits of a filter entry for a table of `slots` entries: the x that share a slot are at most ceil(2^id_bits / slots)
// consecutive values (rg_search_kernel.h: vf_hash), told apart by that many low bits
static uint32_t filter_rem_bits(uint32_t id_bits, uint32_t slots) {
    const uint64_t span = (((uint64_t)1 << id_bits) + slots - 1) / slots;
    uint32_t r = 0;
    while (((uint64_t)1 << r) < span) ++r;
    return r;
}

// visited words of mode 0: slots x ceil(nd / 16) words per context (2.5 MB per slot at 10M nodes).  Grow-only; the slot
// count (and with it the launch grid) is capped by "visited_budget_kb" (default 24 GiB per context), and an allocation
// that fails returns RG_ERR_OOM without touching what the context already had -- the caller then runs the batch in the
// filter + log form, which returns the same bits
// words per slot: ceil(nd / 16) epoch-tagged words, or -- byte form -- nd epoch bytes rounded up to whole 128-byte lines
static uint32_t visited_words(const rg_index *ix, bool bytes) {
    return bytes ? (uint32_t)(((size_t)ix->nd + 127) / 128 * 32) : (ix->nd + 15) / 16;
}
static uint32_t visited_slot_cap(const rg_index *ix, bool bytes) {
    const size_t per = (size_t)visited_words(ix, bytes) * 4;
    return (uint32_t)std::max<size_t>(1, std::min<size_t>(0x7fffffffu, ((size_t)std::max(1, ix->visited_budget_kb) << 10) / std::max<size_t>(per, 1)));
}
static rg_status ensure_visited(rg_index *ix, SearchCtx *cx, uint32_t slots, bool bytes, hipStream_t s) {
    const uint32_t vwords = visited_words(ix, bytes);
    uint32_t *&d_vis = bytes ? cx->d_vtags : cx->d_visited;
    uint32_t *&d_ep = bytes ? cx->d_epoch8 : cx->d_epoch;
    uint32_t &have_slots = bytes ? cx->tslots : cx->slots;
    uint32_t &have_words = bytes ? cx->twords : cx->vwords;
    if (have_slots >= slots && have_words == vwords) return RG_OK;
    uint32_t *nv = nullptr, *ne = nullptr;
    // knob "visited_uncached": the words in memory the L2 does not cache (MTYPE_UC) -- a test then moves a 32-byte sector
    // over the fabric instead of the 128-byte line the L2 fetches for a 4-byte word it will not see again
    // the old buffer goes first: at 10M nodes the tags of a wide-beam launch are 19 GiB, and the new buffer wants the memory
    // (and the classes) the old one held
    const uint32_t old_slots = (d_vis && have_words == vwords) ? have_slots : 0u;
    if (d_vis) { dev_free(d_vis); d_vis = nullptr; have_slots = 0; }
    const bool ok_v = ix->visited_uncached ? hipExtMallocWithFlags(reinterpret_cast<void **>(&nv), (size_t)slots * vwords * 4,
                                                                   ix->visited_uncached == 2 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached) == hipSuccess
                                           : dev_alloc_t(ix->device, (size_t)slots * vwords, &nv) == RG_OK;
    if (ok_v && !ix->visited_uncached && dev_last_plain()) { std::lock_guard<std::mutex> lk(ix->mu); ++ix->n_plain_allocs; }
    dev_trim(ix->device);      // (the allocator's pool of classified granules goes back to the device)
    if (!ok_v || rg_malloc(&ne, (size_t)slots * 4) != hipSuccess) {
        (void)hipGetLastError();
        dev_free(nv);
        // (the context must not lose a working buffer to a failed attempt at a larger one: back to the size it had -- the
        // memory was just released -- so that the launches that fitted before still do)
        if (old_slots && dev_alloc_t(ix->device, (size_t)old_slots * vwords, &nv) == RG_OK) {
            dev_trim(ix->device);
            d_vis = nv;
            (void)hipMemsetAsync(d_vis, 0, (size_t)old_slots * vwords * 4, s);
            if (d_ep) (void)hipMemsetAsync(d_ep, 0, (size_t)old_slots * 4, s);
            have_slots = d_ep ? old_slots : 0;
        }
        return set_error(RG_ERR_OOM, "no room for the visited words of the exact form");
    }
    if (d_ep) (void)hipFree(d_ep);
    if (getenv("RG_TRACE_ALLOC"))   // where an allocation landed (the same launch differs by +- 5 % between two allocations of the tags)
        fprintf(stderr, "[rg_search] visited %s: %u slots x %u words at %p (%.2f GiB)\n", bytes ? "byte tags" : "words", slots, vwords, (void *)nv,
                (double)slots * vwords * 4 / (1u << 30));
    d_vis = nv;
    d_ep = ne;
    have_slots = 0;
    ++cx->allocs;
    // cleared ON THE LAUNCH STREAM: the host-form calls run on a non-blocking private stream, which a fill on the null stream
    // does not hold back -- and recycled memory may hold the tags of an earlier context, whose epochs were the same small numbers
    RG_HIP(hipMemsetAsync(d_vis, 0, (size_t)slots * vwords * 4, s));
    RG_HIP(hipMemsetAsync(d_ep, 0, (size_t)slots * 4, s));
    have_slots = slots;
    have_words = vwords;
    return RG_OK;
}

struct BuildOut { uint2_pod *exp; uint32_t exp_cap, node0; uint32_t *nexp; };
#ifdef RG_K1_PROF
static unsigned long long *g_prof_buf = nullptr;   // [nq][16], set through rg_prof_buffer (instrumented build only)
#endif

// what one K1 launch will look like: kernel form, LDS carve, grid.  Computed in ONE place for the launch itself and for
// rg_search_prepare (which allocates the visited tags of exactly the grid the launch will use).
struct K1Plan {
    K1Launch c;
    int R = 1;
    bool bf = false;
    uint32_t vf_slots = 8;
    bool vbytes = false;      // c.vis == 2 with one epoch byte per node
    uint32_t vs_side = 0;     // c.vis == 3: words of the side table behind the buckets (vf_slots = the buckets' entries then)
    bool ls_front = false;    // c.vis == 2: the same set layout in front of the screen, which is the bl_words behind it
    uint32_t bl_words = 0;
    uint32_t hub_m = 0;       // log2 of the bits of the hub bitmap at the front of the visited region (0 = none)
};

// entries + side-table ids of an exact LDS set cut from `bytes` of filter region (plan_k1 below makes the same split)
static uint32_t lset_capacity_of(uint32_t bytes) {
    const uint32_t side = std::max(16u, bytes / 32u) & ~3u;      // whole buckets of four ids
    return bytes > side * 4u ? (bytes - side * 4u) / 16u * 8u + side : 0u;
}

// mode: 0 exact visited set in HBM, 1 LDS filter, 3 exact set in LDS (lset_need: the nodes it should hold -- resident queries
// are given up, down to ten per CU, until it does)
static rg_status plan_k1(rg_index *ix, int mode, uint32_t nq, uint32_t L, bool with_log, bool build_mode, bool has_qlist, hipStream_t s, K1Plan *out,
                         uint32_t lset_need = 0) {
    const bool bp = build_mode;
    const bool qlist = has_qlist;
    // rows in flight per query: two passes of four pay on graphs with many fresh neighbours per hop (measured: +4 % at
    // out-degree 40, -4 % at 16, where the extra staging only costs resident queries)
    // A batch that leaves most wave slots empty is latency bound per query: LDS is plentiful then, so each query keeps
    // 16 rows in flight (two hop-latency round trips instead of five to ten).
    // The register-staged instantiations (d = 200 / 512) pay for rows in flight with VGPRs, not LDS: 16 rows (4 sets of 16
    // registers) at d = 200, 8 rows (2 sets of 32) at d = 512.
    const int rpp = ix->rows_per_pass > 0 ? ix->rows_per_pass
                    : (dimc_of(ix) == 200) ? 16
                    : (dimc_of(ix) == 512) ? 8
                    : (nq <= (uint32_t)ix->num_cu * 6u && !bp) ? 16
                    : ((double)ix->n_edges >= 28.0 * ix->nd ? 8 : 4);
    int R = std::max(1, std::min(8, rpp / 4));
    if (R == 3) R = 2;
    if (R > 4 && R < 8) R = 4;
    if (R == 8 && !(dimc_of(ix) == 200)) R = 4;
    // opt-in fast mode: plain top-k searches only (never the logging / recount / build launches)
    const bool bf = ix->fast_bf16 && ix->d_base_bf && dimc_of(ix) && !with_log && !bp && !qlist;
    if (bf) R = std::min(R, 2);
    // LDS visited filter, automatic size: the largest of 2^12 .. 2^9 entries that still leaves enough resident queries per
    // CU.  A forgetful filter re-scores nodes (1.37x the distinct ones at L_pq = 500 on the 10M bench index with 2^11
    // entries, 1.28x with 2^12), and every re-scored row is HBM traffic in a kernel that is bandwidth bound.  The
    // register-staged forms keep 16 - 32 rows in flight per query, so eight resident queries still cover the latency:
    // measured on that index (scripts/exp/filter_size_10m.sh, % of 8 TB/s, 2^11 -> 2^12 entries): 81 -> 83.5 at
    // L_pq = 100, 70 -> 75 at 300, 65 -> 70 at 500, 53 -> 56 at 1000; 2^13 loses everywhere (too few queries left).
    // The same holds where the filter only screens the atomics of the exact words (scripts/exp/filter_size_mode0_10m.sh:
    // 60 -> 64 % at L_pq = 500, 54 -> 55 % at 1000 with 2^12 entries).  The LDS-DMA ring forms (4 - 8 rows in flight) keep
    // the older bound of 14.
    const int min_wpc = (dimc_of(ix) && !bf) ? 8 : 14;
    int filter_auto = 9;
    for (int f = 12; f >= 9; --f) {
        filter_auto = f;
        if (f == 9 || (int)(ix->lds_per_cu / search_lds_bytes(ix, L, R, mode, bf, f)) >= min_wpc) break;
    }
    size_t lds = search_lds_bytes(ix, L, R, mode, bf, filter_auto);
    while (lds > ix->lds_per_cu && R > 1) { R >>= 1; lds = search_lds_bytes(ix, L, R, mode, bf, filter_auto); }
    if (lds > ix->lds_per_cu) return set_error(RG_ERR_ARG, "L_pq too large for the 160 KiB LDS of one CU");
    int wpc = (int)std::min<size_t>(ix->lds_per_cu / lds, 32);
    // few resident queries (wide beams): each keeps 32 rows in flight (8 register sets, about 190 VGPRs: 8 waves per CU)
    // The filter forms go there from 12 LDS-limited residents down (L_pq >= 300 at d = 200): eight queries with 32 rows in
    // flight and the LDS of the other three or four in their filters (the fill below) re-read fewer rows than eleven with
    // 16 rows and 2^12 entries -- profiles/r03/k1_ab_box19.jsonl: +1 % at L_pq = 300 - 500, +6 % at 700.
    const int r8_from = (mode != 0 && ix->filter_fill && ix->filter_log2 <= 0 && ix->waves_per_cu <= 0) ? 12 : 8;
    if (ix->rows_per_pass <= 0 && dimc_of(ix) == 200 && !bf && wpc <= r8_from) { R = 8; wpc = std::min(wpc, 8); lds = search_lds_bytes(ix, L, R, mode, bf, filter_auto); }
    // exact-tag form at middling beams (nine to twelve LDS-limited residents, L_pq 300 - 900 at d = 200): eight residents, the LDS of
    // the others goes to the bit screen (more tests of never-marked nodes stay on the CU).  profiles/r04/k1_ab_box17_look_forms.jsonl,
    // % of 8 TB/s at L_pq 300 / 500 / 700: 76.1 / 71.5 / 70.6 as planned before, 76.1 / 73.8 / 71.1 with eight residents
    if (mode == 0 && dimc_of(ix) == 200 && !bf && !bp && ix->waves_per_cu <= 0 && ix->rows_per_pass <= 0 && R == 4 && wpc > 8 && wpc <= 12) wpc = 8;
    // (round 5, with the hub bitmap: between L_pq 250 and 375 -- where LDS would allow 12 - 14 residents with 16 rows in flight -- eight
    // residents with 32 rows each do better: 79.5 - 79.9 against 75.0 - 77.6 % of 8 TB/s at 300, 76.7 - 79.5 against 74.3 - 75.2 at 350, 78.6
    // against 74.9 at 375; from 400 up 16 rows win by a point or two: profiles/r05/k1_ab_box29_rows_in_flight_300_500.txt, k1_ab_box30_*)
    if (mode == 0 && dimc_of(ix) == 200 && !bf && !bp && ix->waves_per_cu <= 0 && ix->rows_per_pass <= 0 && R == 4 && wpc >= 8 && L >= 250u && L <= 375u) {
        R = 8; wpc = 8;
        lds = search_lds_bytes(ix, L, R, mode, bf, filter_auto);
    }
    if (ix->waves_per_cu > 0) wpc = std::min(wpc, ix->waves_per_cu);
    else wpc = std::min(wpc, 24);
    K1Launch c;
    c.R = R; c.vis = mode == 0 ? 0 : mode == 3 ? 3 : 1; c.dimc = dimc_of(ix); c.bf = bf; c.lds = lds;
    // exact words, look-ahead form (rg_search_kernel.h, VIS = 2): the register-staged instantiations over ELL rows that
    // name no node twice; never with the opt-in second expansion, whose two lists share one test phase.  Knob "lookahead":
    // -1 (default) = wherever the form is instantiated; 0 = never; 1 = always; 2 = always, without the early guess of the next
    // adjacency row.  With the word form of its tags (round 3, first version) it won from L_pq 1200 up only (50 - 54 vs 47 - 51 %
    // of 8 TB/s at 2000, 4 - 8 % behind at 300 - 700: more memory instructions per hop); with one epoch BYTE per node -- marks
    // are plain stores, no line is fetched for them -- and the LDS bit screen that spares the tests of never-marked nodes
    // it is the faster form of the exact set at every beam width measured (profiles/r03/k1_ab_box20.jsonl, box21: 68.1 /
    // 65.1 / 61.6 / 58.2 / 54.6 % of 8 TB/s at L_pq 500 / 700 / 1000 / 1500 / 2000 against 60.5 / 57.8 / 54.0 / 49.9 / 46.4 of the
    // returning atomics on the same box).
    // (d = 512, webvid-2.5M shape, profiles/r03/k1_ab_box33_d512.jsonl: 90.3 / 89.3 / 87.5 / 82.2 / 76.9 / 69.5 % at L_pq 50 ... 2000
    // against 86.7 / 86.2 / 84.3 / 80.2 / 76.4 / 67.5 of the returning atomics; d = 200 at L_pq 10 - 100: k1_ab_box28.jsonl)
    const bool look_wanted = ix->lookahead != 0;
    if (mode == 0 && look_wanted && !ix->adj_dups && !ix->multi_expand && !bf && !bp && ix->diag == 0 &&
        ((c.dimc == 200 && R >= 2) || (c.dimc == 512 && (R == 2 || R == 4))))
        c.vis = 2;
    c.gf = gather_form_of(ix, R, bf);
    if (c.vis == 3 && c.gf != 1) return set_error(RG_ERR_ARG, "internal: the exact LDS set exists for the compute-layout instantiations only");
    const bool l2 = ix->metric == RG_METRIC_L2, ell = ix->d_ell != nullptr;
    auto dispatch = [&](const SearchParams &sp) -> rg_status {
        if (l2 && ell) return launch_search_l2_ell(sp, c, s);
        if (l2) return launch_search_l2_csr(sp, c, s);
        if (ell) return launch_search_ip_ell(sp, c, s);
        return launch_search_ip_csr(sp, c, s);
    };
    {   // resident queries per CU: the smaller of what LDS and what the kernel's registers allow
        int occ = 0;
        c.occupancy = &occ;
        SearchParams none{};
        rg_status st = dispatch(none);
        c.occupancy = nullptr;
        if (st != RG_OK) return st;
        if (occ > 0) wpc = std::min(wpc, occ);
    }
    // LDS visited filter, fill (round 3; knob "filter_fill": 1 = default, launches that fill the chip; 2 = every launch; 0 = off): the slot count need not be a power of two
    // (rg_search_kernel.h: vf_hash), so the filter takes the LDS that the resident queries of this launch leave unused --
    // the carve is the last region, the extra entries cost nothing.  The occupancy is asked again with the grown
    // allocation (LDS is handed out in granules) and the filter trimmed until the resident count holds.
    uint32_t vf_slots = (mode != 0 || ix->exact_filter) ? std::max(8u, 1u << filter_log2_of(ix, filter_auto)) : 8u;
    if ((mode != 0 || ix->exact_filter) && ix->filter_fill && ix->filter_log2 <= 0 && wpc >= 1 && ((uint64_t)ix->num_cu * wpc <= nq || ix->filter_fill == 2)) {
        for (;;) {
            const size_t per = (ix->lds_per_cu / (size_t)wpc) / 16 * 16;
            size_t extra = per > lds ? per - lds : 0;
            extra = std::min<size_t>(extra, ((size_t)1 << 16) > (size_t)vf_slots * 2 ? ((size_t)1 << 16) - (size_t)vf_slots * 2 : 0);   // <= 2^15 entries
            for (int tries = 0; extra >= 16 && tries < 8; ++tries) {
                int occ = 0;
                c.occupancy = &occ;
                c.lds = lds + extra;
                SearchParams none{};
                rg_status st = dispatch(none);
                c.occupancy = nullptr;
                if (st != RG_OK) return st;
                if (occ >= wpc) break;
                extra = extra > 512 ? (extra - 512) / 16 * 16 : 0;
                if (tries == 7) extra = 0;
            }
            // the exact LDS set wants room for what a query visits -- and at least 2^(id_bits - 15) buckets, so that a 16-bit entry
            // can tell the ids of one bucket apart: one resident query less, and again, down to eight per CU (narrow beams lose
            // nothing down there: profiles/r04/k1_ab_box13_residents.jsonl, L_pq 40 - 80 at 8 ... 15 residents)
            if (mode == 3 && wpc > 8) {
                const uint32_t bytes = (uint32_t)(vf_slots * 2 + extra), side = std::max(16u, bytes / 32u) & ~3u;
                const uint32_t buckets = bytes > side * 4u ? (bytes - side * 4u) / 16u : 0u;
                if ((lset_need && lset_capacity_of(bytes) < lset_need) || buckets == 0 || filter_rem_bits(id_bits_of(ix->nd), buckets) > 15u) { --wpc; continue; }
            }
            if (extra >= 16) { lds += extra; vf_slots += (uint32_t)(extra / 2); }
            c.lds = lds;
            break;
        }
    }
    c.grid = (uint32_t)std::min<uint64_t>(nq, (uint64_t)ix->num_cu * wpc);
    const bool vbytes = c.vis == 2 && ix->visited_bytes != 0;
    if (mode == 0) c.grid = std::min(c.grid, visited_slot_cap(ix, vbytes));
    // HUB BITS (round 5; rg_search_kernel.h, SearchParams::hub_m): the front of the visited region becomes an exact bitmap of the
    // launch's hubs -- a power of two of bits, at most "hub_pct" percent of the region (default 90; 60 at L_pq <= 420), between 2^10 and 2^19.  The
    // look-ahead tag form takes it (a hub costs no tag line, no tag store and no screen bit).  The exact LDS set does not: a visited hub
    // costs a bitmap about as many bits as a set entry costs, and the form's kernels have no register to spare.
    out->hub_m = 0;
    if (ix->hub_levels && ix->ell_tagged && ix->hub_bits != 0 && !bp && !bf && !qlist && ix->diag == 0 &&
        c.vis == 2) {
        const uint32_t region = vf_slots * 2u;
        // (measured on the 10M bench index, profiles/r05/k1_ab_box1_hub_bits.jsonl, box2: the larger bitmap wins wherever the exact set in
        // front of the tags is not in play -- 75.6 against 74.1 % of 8 TB/s at L_pq 700 with 2^16 instead of 2^15 bits -- and loses
        // a point where it is, L_pq 300 - 400)
        const uint32_t pct = (uint32_t)std::max(1, std::min(95, ix->hub_pct > 0 ? ix->hub_pct : (ix->front_set == 0 || L > 420u) ? 90 : 60));
        uint32_t m = 0;
        if (ix->hub_bits > 0) m = (uint32_t)std::max(8, std::min(19, ix->hub_bits));
        else
            for (uint32_t t = 19; t >= 10; --t)
                if ((uint64_t)(1u << t) / 8u * 100u <= (uint64_t)region * pct) { m = t; break; }
        if (m && (1u << m) / 8u + 64u <= region) {
            out->hub_m = m;
            vf_slots = (region - (1u << m) / 8u) / 2u;
        }
    }
    out->vs_side = 0;
    if (c.vis == 3) {
        // the filter's region becomes the exact set: an eighth of its bytes the side table of full ids, the rest buckets of
        // eight 16-bit entries
        uint32_t bytes = vf_slots * 2u;
        if (ix->lset_bytes > 0) bytes = std::min(bytes, std::max(64u, (uint32_t)ix->lset_bytes / 16u * 16u));   // (tests: a set that queries outgrow)
        const uint32_t side = std::max(16u, bytes / 32u) & ~3u;      // whole buckets of four ids
        const uint32_t buckets = (bytes - side * 4u) / 16u;
        vf_slots = buckets * 8u;
        out->vs_side = side;
        if (buckets == 0 || filter_rem_bits(id_bits_of(ix->nd), buckets) > 15u)
            return set_error(RG_ERR_ARG, "internal: the exact LDS set does not fit this launch");
    }
    out->ls_front = false; out->bl_words = 0;
    if (c.vis == 2 && vbytes && ix->exact_filter && ix->front_set != 0) {
        // look-ahead form with byte tags: the front of the screen's region becomes an exact set (knob "front_set": N = its share of
        // the region in percent; -1, the default = 85 % where that holds at least 0.4 x the nodes a query of this width visits) when
        // the rest still gives the screen 64 words and the set has the 2^(id_bits - 15) buckets its 16-bit entries need.
        // Measured on the 10M bench index (profiles/r04/k1_ab_box38_front_set.txt; % of 8 TB/s, shares 50 / 70 / 85 % against the
        // form without it): L_pq 300 72.7 / 79.5 / 80.9 against 76.1 (85 %: the set holds 0.51 x the visits), 400 75.4 / 75.0 / 74.8
        // against 74.4 (0.38 x), 500 73.6 / 72.0 / 73.2 against 74.1 (0.30 x), 700 71.1 / 70.4 / 68.7 against 71.5: a node the set holds
        // costs no tag store and no tag line when it is met again, but a full set makes every other test walk a bucket and the side
        // table, and what the set takes the screen loses.
        const uint32_t region = vf_slots * 2u;
        const uint32_t pct = (uint32_t)std::min(95, std::max(5, ix->front_set < 0 ? 85 : ix->front_set));
        const uint32_t set_bytes = (uint32_t)((uint64_t)region * pct / 100u) / 16u * 16u;
        const uint32_t side = std::max(16u, set_bytes / 32u) & ~3u;
        const uint32_t buckets = set_bytes > side * 4u ? (set_bytes - side * 4u) / 16u : 0u;
        const uint32_t blw = (region - set_bytes) / 4u;
        bool use = buckets && filter_rem_bits(id_bits_of(ix->nd), buckets) <= 15u && blw >= 64u;
        if (use && ix->front_set < 0) {
            float visits;
            {
                std::lock_guard<std::mutex> lk(ix->mu);
                auto it = ix->evals_at.find(L);
                visits = it != ix->evals_at.end() ? it->second : 44.0f * (float)L;
            }
            use = (float)(buckets * 8u + side) >= 0.4f * visits;
        }
        if (use) {
            out->ls_front = true; out->bl_words = blw; out->vs_side = side;
            vf_slots = buckets * 8u;
        }
    }
    out->c = c; out->R = R; out->bf = bf; out->vf_slots = vf_slots; out->vbytes = vbytes;
    return RG_OK;
}

// one K1 launch.  mode: 0 exact HBM visited words, 1 LDS filter (optionally logging the scored ids).
static rg_status launch_k1(rg_index *ix, SearchCtx *cx, int mode, const float *d_q, uint32_t nq, uint32_t qstride, uint32_t k,
                           uint32_t L, uint32_t *d_ids, float *d_dists, uint32_t *d_cmps, uint32_t *d_hops,
                           const uint32_t *qlist, bool with_log, unsigned long long *d_status, hipStream_t s,
                           const BuildOut *bp = nullptr, uint32_t qbase = 0, unsigned long long *d_totals = nullptr, uint32_t *d_ovf = nullptr,
                           uint32_t lset_need = 0, bool lset_tags = false) {
    K1Plan plan;
    {
        rg_status st = plan_k1(ix, mode, nq, L, with_log, bp != nullptr, qlist != nullptr, s, &plan, lset_need);
        if (st != RG_OK) return st;
    }
    if (mode == 3 && lset_tags) {    // the set's overflow goes to the exact byte tags: one slot of tags per resident query
        plan.vbytes = true;
        plan.c.grid = std::min(plan.c.grid, visited_slot_cap(ix, true));
    }
    const K1Launch &c = plan.c;
    const int R = plan.R;
    if (!qlist && !bp) ix->hub_m_last = plan.hub_m;       // (statistics only)
    const bool bf = plan.bf, vbytes = plan.vbytes;
    const uint32_t vf_slots = plan.vf_slots;
    const size_t lds = c.lds;
    const bool l2 = ix->metric == RG_METRIC_L2, ell = ix->d_ell != nullptr;
    auto dispatch = [&](const SearchParams &sp) -> rg_status {
        if (l2 && ell) return launch_search_l2_ell(sp, c, s);
        if (l2) return launch_search_l2_csr(sp, c, s);
        if (ell) return launch_search_ip_ell(sp, c, s);
        return launch_search_ip_csr(sp, c, s);
    };
    if (mode == 0 || (mode == 3 && lset_tags)) {
        rg_status st = ensure_visited(ix, cx, c.grid, vbytes, s);
        if (st != RG_OK) return st;
    }
    RG_HIP(hipMemsetAsync(cx->d_counter, 0, 4, s));
    SearchParams P;
    P.base = ix->d_base; P.stride = ix->stride; P.dim = ix->dim; P.nd = ix->nd;
    P.ell = ix->d_ell; P.ell_stride = ix->ell_stride; P.offsets = ix->d_offsets; P.nbrs = ix->d_nbrs;
    P.ep = ix->ep; P.queries = d_q; P.nq = nq; P.qstride = qstride; P.k = k; P.L = L;
    P.out_ids = d_ids; P.out_dists = d_dists; P.out_cmps = d_cmps; P.out_hops = d_hops;
    P.visited = (mode == 0 || (mode == 3 && lset_tags)) ? (vbytes ? cx->d_vtags : cx->d_visited) : nullptr;
    P.vwords = vbytes ? cx->twords : cx->vwords; P.slot_epoch = vbytes ? cx->d_epoch8 : cx->d_epoch;
    P.vbytes = vbytes ? 1u : 0u;
    P.roll = ix->gather_roll ? 1u : 0u;
    P.counter = cx->d_counter; P.status = d_status;
    P.stage_floats = (uint32_t)stage_pass_floats(ix, bf);
    P.stage_total = (uint32_t)stage_total_floats(ix, R, bf);
    P.base_bf = bf ? ix->d_base_bf : nullptr; P.stride_bf = ix->stride_bf;
    // remainder block of a row in the register-staged gather (d = 200: elements 192..199): behind the row, or -- split
    // rows -- in the per-edge tail array
    P.tail_base = ix->d_base + (ix->dim / 64u) * 64u; P.tail_stride = ix->stride; P.tail_off = nullptr; P.ep_tail = ix->ep;
    if (ix->d_tail_off && ix->split_rows && ix->d_ell && !bp && !bf && dimc_of(ix) == 200) {
        P.base = ix->d_main; P.stride = ix->main_dim;
        P.tail_base = ix->d_etail; P.tail_stride = ix->tail_dim; P.tail_off = ix->d_tail_off; P.ep_tail = (uint32_t)ix->n_edges;
    }
    P.diag = (uint32_t)ix->diag;
    P.qbase = qbase;
    P.vf_slots = vf_slots;
    P.vf_rem_bits = filter_rem_bits(id_bits_of(ix->nd), (c.vis == 3 || plan.ls_front) ? vf_slots / 8u : vf_slots);
    P.ls_front = plan.ls_front ? 1u : 0u; P.bl_words = plan.bl_words;
    P.hub_m = plan.hub_m; P.hub_words = plan.hub_m ? (1u << plan.hub_m) / 32u : 0u;
    P.vs_side = plan.vs_side;
    P.ovf_count = d_ovf; P.ovf_list = d_ovf ? d_ovf + 2 : nullptr;       // (the batch record's overflow list: count, K4 work counter, queries)
    P.lset_left = d_totals ? d_totals + 2 : nullptr;
    if (c.vis == 3 && (!d_ovf || !d_totals || !with_log)) return set_error(RG_ERR_ARG, "internal: the exact LDS set needs the batch's totals, overflow list and logs");
    P.vf_front = (mode == 0 && ix->exact_filter) ? 1u : 0u;
    P.id_bits = id_bits_of(ix->nd);
    P.qlog = with_log ? cx->d_qlog : nullptr; P.logcap = cx->logcap; P.qlog_n = with_log ? cx->d_qlog_n : nullptr;
    P.qlist = qlist;
    // opt-in, NOT parity: two expansions per iteration (never in the build-mode searches, whose expansion lists must be the
    // reference's)
    P.spec = (ix->multi_expand && !bp) ? 2u : 0u;
    P.look = ix->lookahead == 2 ? 0u : 1u;
    P.log_early = ix->log_early ? 1u : 0u;
    P.front_scores = nullptr; P.front_stride = 0;
    if (cx->cur_front && !qlist && !bp && !bf) { P.front_scores = cx->cur_front + (size_t)qbase * cx->front_stride; P.front_stride = cx->front_stride; }
    // In-kernel exact distinct count (rg_search_kernel.h: wave_distinct_half): beams up to "count_in_k1" wide (default 40)
    // log a thousand or two ids per query, which the wave counts itself at the end of the query; K4 then finds nothing to
    // do.  Measured on the 10M bench index (profiles/r03/k1_ab_box11.jsonl, % of 8 TB/s, K4 -> in-kernel): 81.6 -> 83.4 at
    // L_pq = 20 (the sweep's 10 - 40 points gain 3 - 4 %: a separate launch and its tail weigh most where a step takes a
    // millisecond or two), level at 50 - 60, 1 - 2 % behind from 80 up (one wave's CAS chains against K4's full workgroups).
    // The table takes the LDS from the merge scratch on (beam, log line, filter): the largest power of two of words whose
    // buckets + side table fit; remainders must fit 15 bits (indexes of up to 2^(tbits + 13) nodes).
    P.count_tbits = 0;
    P.count_mode = 0;
    P.totals = d_totals;
    const bool count_ok = with_log && d_totals && !qlist && !bp && ix->count_table_auto && !ix->count_full_ids && ix->log_cap_knob <= 0;
    const bool inline_count = count_ok && mode != 3 && ix->count_in_k1 != 0 && L <= (uint32_t)(ix->count_in_k1 < 0 ? 40 : ix->count_in_k1);
    const bool lset_count = mode == 3 && with_log && !ix->count_full_ids;      // a query that outgrows the exact set counts its own short log
    if (inline_count || lset_count) {
        const size_t fixed = (size_t)P.stage_total * 4 + (dimc_of(ix) ? 0 : (size_t)ix->dim * 4) + 2 * kCand * 4;   // in front of the merge scratch
        const size_t region = lds > fixed ? lds - fixed : 0;
        uint32_t tb = 0;
        for (uint32_t t = 8; t <= 13; ++t)
            if (((size_t)4 << t) + ((size_t)4 << (t - 3)) <= region) tb = t;
        if (tb >= 8 && id_bits_of(ix->nd) <= tb - 2 + 15) { P.count_tbits = tb; P.count_mode = mode == 3 ? 0u : 1u; }
    }
    P.id_mask = (ix->ell_tagged && !bp) ? 0x00ffffffu : 0xffffffffu;
    // (the in-degree tag of a neighbour word is a nibble since round 5 -- min(15, in-degree): a knob above 15 means 15)
    P.vf_min_indeg = (uint32_t)std::max(0, std::min(15, ix->filter_min_indeg));
#ifdef RG_K1_PROF
    P.prof = (qlist || bp) ? nullptr : g_prof_buf;
#endif
    P.out_exp = nullptr; P.exp_cap = 0; P.tgt_base = 0; P.out_nexp = nullptr;
    if (bp) { P.out_exp = reinterpret_cast<uint2 *>(bp->exp); P.exp_cap = bp->exp_cap; P.tgt_base = bp->node0; P.out_nexp = bp->nexp; }
    return dispatch(P);
}

static rg_status ensure_qlog(rg_index *ix, SearchCtx *cx, uint32_t nq) {
    // per-query id log: 128K ids (512 KiB) each; a batch larger than the log budget allows is searched in sub-batches
    // that reuse the same logs (search_dev).  Longer logs take the exact fallback pass.
    uint32_t cap = 1u << 17;
    if (ix->log_cap_knob > 0) cap = (uint32_t)ix->log_cap_knob;
    const size_t budget = (size_t)std::max(1, ix->log_budget_kb) << 10;
    const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(nq, budget / ((size_t)cap * 4)));
    if (cx->d_qlog && cx->qlog_nq >= chunk && cx->logcap == cap) { cx->qlog_chunk = chunk; return RG_OK; }
    dev_free(cx->d_qlog);
    if (cx->d_qlog_n) (void)hipFree(cx->d_qlog_n);
    cx->d_qlog = cx->d_qlog_n = nullptr;
    cx->qlog_nq = 0;
    ++cx->allocs;
    {
        rg_status as = dev_alloc_t(ix->device, (size_t)chunk * cap, &cx->d_qlog);
        if (as == RG_OK && dev_last_plain()) { std::lock_guard<std::mutex> lk(ix->mu); ++ix->n_plain_allocs; }
        dev_trim(ix->device);
        if (as != RG_OK) return as;
    }
    RG_HIP(rg_malloc(&cx->d_qlog_n, (size_t)chunk * 4));
    cx->qlog_nq = chunk;
    cx->qlog_chunk = chunk;
    cx->logcap = cap;
    return RG_OK;
}

static rg_status finish_batches(rg_index *ix, SearchCtx *cx, hipStream_t s, uint32_t k);

// enqueue one batch on `s` (no synchronisation); the batch record joins cx->pending
static rg_status search_dev(rg_index *ix, SearchCtx *cx, const float *d_q, uint32_t nq, uint32_t qstride, uint32_t k, uint32_t L,
                            uint32_t *d_ids, float *d_dists, uint32_t *d_cmps, uint32_t *d_hops, hipStream_t s) {
    // a caller that never waits must not pile up batch records: collect the finished ones first (errors stay deferred)
    if (cx->pending.size() >= 32) {
        rg_status st = finish_batches(ix, cx, s, 0);
        if (st != RG_OK && st != RG_ERR_NOT_ENOUGH) return st;
        if (st == RG_ERR_NOT_ENOUGH && cx->deferred == RG_OK) {   // reported by the next rg_search_wait on this stream
            cx->deferred = st;
            cx->deferred_msg = rg_last_error();
        }
    }
    Batch *b = nullptr;
    rg_status st = take_batch(cx, nq, &b);
    if (st != RG_OK) return st;
    cx->log_holds = 0;
    auto fail = [&](rg_status e) { cx->spare.push_back(b); return e; };
    b->q = d_q; b->nq = nq; b->qstride = qstride; b->k = k; b->L = L;
    b->ids = d_ids; b->dists = d_dists; b->cmps = d_cmps; b->hops = d_hops;
    if (hipMemsetAsync(b->d_stat, 0xff, 8, s) != hipSuccess || hipMemsetAsync(b->d_stat + 1, 0, 24, s) != hipSuccess)
        return fail(set_error(RG_ERR_DEVICE, "hipMemsetAsync failed"));
    const bool fast = ix->fast_bf16 && ix->d_base_bf && dimc_of(ix);
    const bool exact_count = ix->visited_mode == 2 && d_cmps != nullptr && !fast;
    cx->cur_front = nullptr;
    if (ix->shared_frontier && ix->d_front_ids && ix->front_n > 1 && !fast) {
        // the first hop of every query of the batch scores the same rows: once for the batch, with the exact routine
        const uint32_t fs = (ix->front_n + 3u) & ~3u;
        if (cx->front_cap < (size_t)nq * fs) {
            if (cx->d_front) (void)hipFree(cx->d_front);
            cx->d_front = nullptr; cx->front_cap = 0;
            if (rg_malloc(&cx->d_front, (size_t)nq * fs * 4) != hipSuccess) return fail(set_error(RG_ERR_OOM, "no room for the shared-frontier scores"));
            cx->front_cap = (size_t)nq * fs;
        }
        constexpr int FR = 4;
        const uint32_t stage_floats = ((ix->dim + 63) / 64) * 256;
        const size_t flds = (size_t)FR * stage_floats * 4 + (size_t)ix->dim * 4;
        const uint32_t fgrid = std::min<uint32_t>(nq, (uint32_t)ix->num_cu * 16u);
        if (ix->metric == RG_METRIC_L2)
            hipLaunchKernelGGL((rg_front_score_kernel<true, FR>), dim3(fgrid), dim3(64), flds, s, ix->d_base, ix->stride, ix->dim, d_q, nq, qstride,
                               ix->d_front_ids, ix->front_n, cx->d_front, fs, stage_floats);
        else
            hipLaunchKernelGGL((rg_front_score_kernel<false, FR>), dim3(fgrid), dim3(64), flds, s, ix->d_base, ix->stride, ix->dim, d_q, nq, qstride,
                               ix->d_front_ids, ix->front_n, cx->d_front, fs, stage_floats);
        if (hipGetLastError() != hipSuccess) return fail(set_error(RG_ERR_DEVICE, "shared-frontier launch failed"));
        cx->cur_front = cx->d_front;
        cx->front_stride = fs;
    }
    uint32_t exact_from_L, trial_L;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        exact_from_L = ix->exact_from_L;
        trial_L = ix->trial_L;
    }
    const uint32_t allocs0 = cx->allocs;
    auto done = [&]() -> rg_status {
        b->h_stat[4] = 0;
        b->cold = cx->allocs != allocs0;
        if (hipMemcpyAsync(b->h_stat, b->d_stat, 32, hipMemcpyDeviceToHost, s) != hipSuccess)      // status, two totals, queries that left the exact LDS set
            return fail(set_error(RG_ERR_DEVICE, "hipMemcpyAsync failed"));
        if (b->counted && hipMemcpyAsync(b->h_stat + 4, b->d_ovf, 4, hipMemcpyDeviceToHost, s) != hipSuccess)
            return fail(set_error(RG_ERR_DEVICE, "hipMemcpyAsync failed"));
        std::lock_guard<std::mutex> lk(ix->mu);
        cx->pending.push_back(b);
        ++(b->mode == 3 ? ix->n_batches_lset : b->mode == 0 ? ix->n_batches_exact_hbm : b->counted ? ix->n_batches_filter_log : ix->n_batches_filter_only);
        return RG_OK;
    };
    // fast mode under the default visited mode: the exact words from the beam width on at which the parity batches (if
    // there were any) found them faster, the LDS filter alone below it; "visited" 0 / 1 force one or the other
    if (fast && ix->visited_mode == 2 && L >= exact_from_L) {
        st = launch_k1(ix, cx, 0, d_q, nq, qstride, k, L, d_ids, d_dists, d_cmps, d_hops, nullptr, false, b->d_stat, s);
        return st == RG_OK ? done() : fail(st);
    }
    // Narrow and middling beams (round 4): the exact visited set in LDS (K1 VIS = 3).  What a query visits -- a few thousand nodes
    // at L_pq <= 100 -- fits the LDS region the forgetful filter has, as eight-entry buckets of 16-bit remainders (K4's set, kept by
    // the searching wave itself): nothing is scored twice, so cmps is exact as counted -- no id log to store, no K4 behind the
    // launch, no de-duplicating inserts.  The launch gives up resident queries, down to eight per CU, until the set holds 1.75 x the
    // nodes a query of this width visits (the mean of the last counted batch; 44 x L_pq before there is one).  A query that
    // outgrows its set finishes in the forgetful form and counts its own -- short -- log (same bits), and that is cheap enough
    // for the form to stay ahead while the set still holds 0.8 x the mean visits: % of 8 TB/s on the 10M bench index, this form /
    // filter + log + K4 / exact tags (profiles/r04/k1_ab_box21_lset_forced_wide.jsonl): L_pq 100 87.7 / 82.9 / 79.8, 150 85.1 /
    // 80.6 / 79.0 (every query outgrows), 200 82.4 / 78.0 / 75.3, 300 74.3 / 77.1 / 77.1, 500 62.5 / 73.9 / 73.0.
    if (exact_count && ix->lset != 0 && ix->filter_log2 <= 0 && ix->log_cap_knob <= 0 && !ix->multi_expand && ix->diag == 0 && dimc_of(ix)) {
        uint32_t need;
        {
            std::lock_guard<std::mutex> lk(ix->mu);
            auto it = ix->evals_at.find(L);
            // (1.75: between 1.63 and 1.73 x the mean the form falls off a cliff -- 81.6 vs 87.8 % of 8 TB/s at L_pq = 60 with twelve
            // and eleven residents, profiles/r04/k1_ab_box14_lset_plan.txt: full buckets send their nodes to the side table,
            // whose linear probes are CAS round trips)
            need = (uint32_t)(1.75f * (it != ix->evals_at.end() ? it->second : 44.0f * (float)L));
        }
        const bool forced = ix->lset > 0 && L <= (uint32_t)ix->lset;
        // knob "lset_tags": where the set alone does not pay, the nodes it has no room for go to the exact byte tags in HBM
        // (same kernel, P.visited set): exact without a log at any width whose beam leaves room for 512 buckets
        // Measured on the 10M bench index (profiles/r04/k1_ab_box35_lset_tags.txt, % of 8 TB/s, this form / the default forms): L_pq 240
        // 81.2 / 77.4 (the set holds 0.74 x the mean visits), 300 75.7 / 75.1 (0.60 x), 400 71.7 / 75.2 (0.45 x), 500 68.6 / 72.5, 1000 60.2 /
        // 67.1 -- the tests behind the set are a dependent round trip per hop, which the look-ahead form of the tags hides; so the form is
        // used (knob 1, default) where the set holds at least 0.6 x the mean visits, and everywhere with knob 2 (tests)
        const bool tags_ok = ix->lset_tags > 0 && !forced;
        if (forced || (ix->lset < 0 && (L <= 512u || ix->lset_tags >= 2))) {
            K1Plan plan;
            rg_status ps = plan_k1(ix, 3, nq, L, true, false, false, s, &plan, need);
            {
                static const bool trace = getenv("RG_TRACE_ADAPTIVE") != nullptr;
                if (trace)
                    fprintf(stderr, "[rg_search] exact LDS set at L=%u: need %u, plan %s, holds %u (grid %u = %u per CU, R=%d)\n", L, need, ps == RG_OK ? "ok" : rg_last_error(),
                            ps == RG_OK ? plan.vf_slots + plan.vs_side : 0u, plan.c.grid, plan.c.grid / (uint32_t)std::max(1, ix->num_cu), plan.R);
            }
            // (holds >= 0.8 x the mean visits = 0.457 x need)
            const bool pure = forced || (L <= 512u && (double)(plan.vf_slots + plan.vs_side) >= 0.457 * (double)need);
            // (never over rows that name a node twice: two lanes of one hop that bring the same node and find no room in the set would both
            // read a stale tag with a plain load and both call the node fresh -- the pure form settles them with its CAS or, once it
            // logs, with de-duplicating inserts; the look-ahead form is kept off such rows for the same reason)
            const bool with_tags = !pure && tags_ok && !ix->adj_dups && (ix->lset_tags >= 2 || (L <= 512u && (double)(plan.vf_slots + plan.vs_side) >= 0.343 * (double)need));      // (round 5: 0.64 x instead of 0.6 x the visits tried against the look-ahead tags with hubs -- within the box-to-box noise, profiles/r05/k1_ab_box33_*, k1_ab_box34_*)
            if (ps == RG_OK && (pure || with_tags) && (st = ensure_qlog(ix, cx, nq)) == RG_OK && nq <= cx->qlog_chunk) {
                if (hipMemsetAsync(b->d_ovf, 0, 8, s) != hipSuccess) return fail(set_error(RG_ERR_DEVICE, "hipMemsetAsync failed"));
                b->mode = 3;
                st = launch_k1(ix, cx, 3, d_q, nq, qstride, k, L, d_ids, d_dists, d_cmps, d_hops, nullptr, true, b->d_stat, s, nullptr, 0, b->d_stat + 1,
                               b->d_ovf, need, with_tags);
                if (st == RG_ERR_OOM && with_tags) { st = RG_OK; ps = RG_ERR_OOM; b->mode = 2; }     // no room for the tags: the forms below
                else {
                if (st != RG_OK) return fail(st);
                cx->log_holds = 0;
                b->counted = true;
                return done();
                }
            }
            if (st != RG_OK) return fail(st);
        }
    }
    // Adaptive default: both exact forms return the same bits.  When a batch showed the LDS filter re-scoring nodes
    // at this beam width (performed > 1.04 x distinct: long searches on indexes with locality), the next batch
    // of that width runs on the exact HBM words as a timed trial; the faster form is kept from that width on (which one
    // wins depends on the index: the words of a 10M-node index are 10 GB of random atomics, those of a 2M-node index
    // mostly cache resident -- scripts/exp/visited_modes_real.py).
    if (exact_count && ix->filter_log2 <= 0 && ix->adaptive) {
        if (!b->ev0 && (hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess))
            return fail(set_error(RG_ERR_DEVICE, "hipEventCreate failed"));
        const bool trial = trial_L == L && nq >= 1000;
        b->timed = true;
        b->mode = (L >= exact_from_L || trial) ? 0 : 2;
        b->is_trial = trial && L < exact_from_L;
        (void)hipEventRecord(b->ev0, s);
        if (b->mode == 0) {
            st = launch_k1(ix, cx, 0, d_q, nq, qstride, k, L, d_ids, d_dists, d_cmps, d_hops, nullptr, false, b->d_stat, s);
            if (st == RG_OK) {
                (void)hipEventRecord(b->ev1, s);
                return done();
            }
            if (st != RG_ERR_OOM) return fail(st);
            b->mode = 2; b->timed = false; b->is_trial = false;   // no room for the words: the other exact form, same bits
            {   // ... and remembered: the trial of this width is over (a multi-GiB hipMalloc that fails is not retried every batch)
                std::lock_guard<std::mutex> lk(ix->mu);
                ix->filter_ok_upto = std::max(ix->filter_ok_upto, L);
                if (ix->trial_L == L) ix->trial_L = 0;
                if (ix->exact_from_L <= L) ix->exact_from_L = 0xffffffffu;
            }
        }
    }
    if (!exact_count) {
        b->mode = ix->visited_mode == 0 ? 0 : 2;
        st = launch_k1(ix, cx, ix->visited_mode == 0 ? 0 : 1, d_q, nq, qstride, k, L, d_ids, d_dists, d_cmps, d_hops, nullptr, false,
                       b->d_stat, s);
        return st == RG_OK ? done() : fail(st);
    }
    // mode 2: LDS-filter search with id log, then the exact distinct count (K4); overflowed logs are re-counted by an
    // exact pass inside rg_search_wait
    st = ensure_qlog(ix, cx, nq);
    if (st != RG_OK) return fail(st);
    if (hipMemsetAsync(b->d_ovf, 0, 8, s) != hipSuccess) return fail(set_error(RG_ERR_DEVICE, "hipMemsetAsync failed"));
    // table: up to 2^15 words = 128 KiB of LDS (+ 16 KiB side table in the half-word form).  Narrow beams log a few
    // thousand ids per query; a table sized for them (L_pq x 64 ids) is cleared in a fraction of the time -- clearing is
    // what K4 costs at small L_pq (a tenth of the whole step at L_pq = 50 with the full table) -- and two workgroups fit
    // a CU.  A log that outgrows its table is counted in hash partitions, as before.
    uint32_t tbits = (uint32_t)std::max(6, std::min(15, ix->count_table_log2));
    if (ix->count_table_auto)
        while (tbits > 11 && (uint64_t)((1u << (tbits - 1)) / 4u) * 5u >= (uint64_t)L * 64u) --tbits;
    const uint32_t bbits = tbits - 2u;
    const uint32_t id_bits = std::max(id_bits_of(ix->nd), bbits + 1u);
    const bool half = id_bits - bbits <= 15u && !ix->count_full_ids;
    const size_t lds = half ? ((size_t)4 << tbits) + ((size_t)4 << (tbits - 3u)) : (size_t)4 << tbits;
    for (uint32_t q0 = 0; q0 < nq; q0 += cx->qlog_chunk) {
        const uint32_t nqc = std::min(cx->qlog_chunk, nq - q0);
        if (q0) (void)hipMemsetAsync(b->d_ovf + 1, 0, 4, s);   // K4 work counter; the overflow count keeps running
        st = launch_k1(ix, cx, 1, d_q + (size_t)q0 * qstride, nqc, qstride, k, L, d_ids ? d_ids + (size_t)q0 * k : nullptr,
                       d_dists ? d_dists + (size_t)q0 * k : nullptr, d_cmps + q0, d_hops ? d_hops + q0 : nullptr, nullptr, true,
                       b->d_stat, s, nullptr, q0, b->d_stat + 1);
        if (st != RG_OK) return fail(st);
        // workgroup width by table size: a few thousand ids per query want many small workgroups per CU (a 1024-thread
        // group spends its time in barriers), the full table wants the 16 waves that cover its LDS latency
        const uint32_t k4_threads = tbits <= 13u ? 256u : 1024u;
        const uint32_t k4_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(2048u / k4_threads, ix->lds_per_cu / (lds + 2048)));
        const dim3 grid(std::min<uint32_t>(nqc, (uint32_t)ix->num_cu * k4_per_cu));
        if (half) {
            auto kern = rg_distinct_kernel<true>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, grid, dim3(k4_threads), lds, s, cx->d_qlog, cx->logcap, cx->d_qlog_n, nqc, d_cmps + q0, b->d_ovf + 2,
                               b->d_ovf, b->d_ovf + 1, tbits, id_bits, q0, b->d_stat + 1);
        } else {
            auto kern = rg_distinct_kernel<false>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, grid, dim3(k4_threads), lds, s, cx->d_qlog, cx->logcap, cx->d_qlog_n, nqc, d_cmps + q0, b->d_ovf + 2,
                               b->d_ovf, b->d_ovf + 1, tbits, id_bits, q0, b->d_stat + 1);
        }
    }
    if (hipGetLastError() != hipSuccess) return fail(set_error(RG_ERR_DEVICE, "K4 launch failed"));
    if (b->timed) (void)hipEventRecord(b->ev1, s);
    cx->log_holds = nq <= cx->qlog_chunk ? nq : 0;   // rg_search_reuse_stats: the logs of a batch searched in one piece
    b->counted = true;
    return done();
}

// synchronise `s` and finish every batch pending on the context, in order: adaptive-mode bookkeeping, the exact recount
// of overflowed id logs, and the first deferred "not enough results" (index_bipartite.cpp:2408-2412)
static rg_status finish_batches(rg_index *ix, SearchCtx *cx, hipStream_t s, uint32_t k) {
    trace_stale("finish_batches (entry: left by the enqueue)");
    std::vector<Batch *> todo;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        todo.swap(cx->pending);
    }
    RG_HIP(hipStreamSynchronize(s));
    rg_status first = RG_OK;
    std::string first_msg;
    for (Batch *b : todo) {
        const unsigned long long v = b->h_stat[0];
        float per_q = 0.0f;   // time per query of the batch (adaptive default only)
        if (b->timed) {
            float ms = 0.0f;
            // a cold batch (its enqueue allocated the id logs or the visited words between the two events) is not a
            // measurement: no verdict, no trial request; the next batch of this width decides
            if (!b->cold && hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess && b->nq) per_q = ms / (float)b->nq;
            std::lock_guard<std::mutex> lk(ix->mu);
            static const bool trace = getenv("RG_TRACE_ADAPTIVE") != nullptr;   // decisions of the adaptive default on stderr
            if (trace)
                fprintf(stderr, "[rg_search] batch L=%u nq=%u form=%s%s%s: %.3f us/query\n", b->L, b->nq, b->mode == 0 ? "exact words" : "filter+log",
                        b->is_trial ? " (trial)" : "", b->cold ? " (cold: not a measurement)" : "", per_q * 1e3f);
            if (b->mode == 0 && b->is_trial && per_q > 0.0f) {   // verdict of the trial
                // (round 4: any measurable win -- with the tags spread over the memory classes the exact set is 2 - 13 % ahead from
                // L_pq 500 up on the 10M bench index, and a verdict that asked for 3 % left the 500 point on the slower form)
                if (per_q < 0.99f * ix->filter_per_q) ix->exact_from_L = std::min(ix->exact_from_L, b->L);
                else ix->filter_ok_upto = std::max(ix->filter_ok_upto, b->L);
                if (ix->trial_L == b->L) ix->trial_L = 0;
                if (trace)
                    fprintf(stderr, "[rg_search]   verdict at L=%u: exact words %.3f vs filter+log %.3f us/query -> exact from L=%u, filter kept up to L=%u\n",
                            b->L, per_q * 1e3f, ix->filter_per_q * 1e3f, ix->exact_from_L, ix->filter_ok_upto);
            }
        }
        if (b->counted) {
            const unsigned long long performed = b->h_stat[1], distinct = b->h_stat[2];
            {
                std::lock_guard<std::mutex> lk(ix->mu);
                if (distinct > 0 && b->nq >= 64) ix->evals_at[b->L] = (float)((double)distinct / (double)b->nq);      // what the exact LDS set is sized by
                if (b->mode == 3) {
                    const unsigned long long left = b->h_stat[3];
                    ix->n_lset_left += left;
                    static const bool trace = getenv("RG_TRACE_ADAPTIVE") != nullptr;
                    if (trace) fprintf(stderr, "[rg_search] batch L=%u nq=%u form=exact LDS set: %llu queries outgrew it\n", b->L, b->nq, left);
                }
                // (round 4: from 4 % of re-scored nodes; round 3: 8 % -- it was 30 % -- : with byte tags and the bit screen the exact set wins
                // earlier, at d = 512 from L_pq 200 where the filter re-scores a sixth; a trial costs one batch in the other form)
                if (distinct > 0 && (double)performed > 1.04 * (double)distinct && per_q > 0.0f && b->nq >= 1000 &&
                    b->L > ix->filter_ok_upto && b->L < ix->exact_from_L) {
                    ix->trial_L = b->L;      // next batch of this width: the exact words, timed
                    ix->filter_per_q = per_q;
                }
            }
            const uint32_t novf = (uint32_t)(b->h_stat[4] & 0xffffffffu);
            if (novf > 0) { std::lock_guard<std::mutex> lk(ix->mu); ix->n_recounted += novf; }
            if (novf > 0 && v == ~0ull && first == RG_OK) {
                // logs that did not fit: recount those queries with the exact HBM visited words (only cmps is rewritten)
                rg_status st = launch_k1(ix, cx, 0, b->q, novf, b->qstride, b->k, b->L, b->ids, b->dists, b->cmps, b->hops, b->d_ovf + 2,
                                         false, cx->d_scratch_stat, s);
                if (st == RG_OK && hipStreamSynchronize(s) != hipSuccess) st = set_error(RG_ERR_DEVICE, "recount pass failed");
                if (st != RG_OK) { first = st; first_msg = rg_last_error(); }
            }
        }
        if (v != ~0ull && first == RG_OK) {
            char buf[160];
            const uint32_t kk = k ? k : b->k;
            snprintf(buf, sizeof buf, "not enough results: %u, expected: %u (query %u)", (unsigned)(v & 0xffffffffu), kk, (unsigned)(v >> 32));
            first = RG_ERR_NOT_ENOUGH;
            first_msg = buf;
        }
    }
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        for (Batch *b : todo) cx->spare.push_back(b);
    }
    if (first != RG_OK) return set_error(first, first_msg);
    return RG_OK;
}

static rg_status check_search_args(const rg_index *ix, uint32_t qstride, uint32_t k, uint32_t L) {
    if (!ix) return set_error(RG_ERR_ARG, "null index");
    if (k > L) return set_error(RG_ERR_ARG, "L_pq must greater or equal than k");  // test_search_roargraph.cpp:192-195
    if (k == 0 || L == 0) return set_error(RG_ERR_ARG, "k and L_pq must be positive");
    if (qstride < ix->dim) return set_error(RG_ERR_ARG, "query stride smaller than the index dimension");
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
    // the build drives one stream from one thread: its context stays keyed to that stream for the life of the index
    SearchCtx *cx = nullptr;
    rg_status st = acquire_ctx(ix, (hipStream_t)stream, false, &cx);
    if (st != RG_OK) return st;
    BuildOut bo{d_exp, exp_cap, node0, d_nexp};
    // queries are the base rows themselves; k = 1 (no top-k is written in build mode)
    return launch_k1(ix, cx, 1, ix->d_base + (size_t)node0 * ix->stride, n, ix->stride, 1, L, nullptr, nullptr, nullptr, nullptr,
                     nullptr, false, cx->d_scratch_stat, (hipStream_t)stream, &bo);
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
    hipError_t e = rg_malloc(&ix->d_ell, (size_t)nd * ell_stride * 4);
    if (e == hipSuccess) e = hipMemset(ix->d_ell, 0, (size_t)nd * ell_stride * 4);
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

// rows [idx[r]] of the ELL array <- rows[r] (the build's graph snapshot goes up as the rows that changed since the last one)
__global__ void rg_ell_scatter_kernel(uint32_t *ell, const uint32_t *rows, const uint32_t *idx, uint32_t n, uint32_t S) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * S) return;
    const uint32_t r = (uint32_t)(i / S), c = (uint32_t)(i % S);
    ell[(size_t)idx[r] * S + c] = rows[i];
}
rg_status build_index_update_rows(rg_index *ix, const uint32_t *d_rows, const uint32_t *d_idx, uint32_t n, void *stream) {
    if (n == 0) return RG_OK;
    RG_HIP(hipSetDevice(ix->device));
    const size_t words = (size_t)n * ix->ell_stride;
    hipLaunchKernelGGL(rg_ell_scatter_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ix->d_ell, d_rows, d_idx, n,
                       ix->ell_stride);
    RG_HIP(hipGetLastError());
    return RG_OK;
}

// host-form search in two halves (rg_search: one index; rg_search_sharded: one per replica, all in flight together)
struct HostSearch {
    rg_index *ix = nullptr;
    SearchCtx *cx = nullptr;
    uint32_t lo = 0, n = 0;
};

static rg_status host_search_begin(rg_index *ix, const float *hq, uint32_t n, uint32_t d, uint32_t k, uint32_t L, HostSearch *hs) {
    hs->ix = ix; hs->n = n;
    if (n == 0) return RG_OK;
    RG_HIP(hipSetDevice(ix->device));
    rg_status st = acquire_ctx(ix, nullptr, true, &hs->cx);
    if (st != RG_OK) return st;
    SearchCtx *cx = hs->cx;
    const size_t qn = (size_t)n * d, rn = (size_t)n * k, cn = (size_t)n * 2;
    if (cx->q_cap < qn) { if (cx->d_q) (void)hipFree(cx->d_q); cx->d_q = nullptr; cx->q_cap = 0; RG_HIP(rg_malloc(&cx->d_q, qn * 4)); cx->q_cap = qn; }
    if (cx->res_cap < rn) {
        if (cx->d_ids) (void)hipFree(cx->d_ids);
        if (cx->d_dist) (void)hipFree(cx->d_dist);
        cx->d_ids = nullptr; cx->d_dist = nullptr; cx->res_cap = 0;
        RG_HIP(rg_malloc(&cx->d_ids, rn * 4));
        RG_HIP(rg_malloc(&cx->d_dist, rn * 4));
        cx->res_cap = rn;
    }
    if (cx->ch_cap < cn) { if (cx->d_ch) (void)hipFree(cx->d_ch); cx->d_ch = nullptr; cx->ch_cap = 0; RG_HIP(rg_malloc(&cx->d_ch, cn * 4)); cx->ch_cap = cn; }
    hipStream_t s = cx->own;
    // Pinned staging (round 3): the caller's buffers are pageable, and a pageable hipMemcpy stages through the runtime's own
    // bounce buffers one chunk at a time on the calling thread.  The queries are copied into this context's pinned buffer
    // by a few threads and leave with ONE asynchronous DMA; the results come back into pinned memory the same way, the
    // whole call synchronises once.
    const size_t hb = std::max(qn * 4, (rn * 2 + cn) * 4);
    if (cx->h_cap < hb) {
        if (cx->h_pin) (void)hipHostFree(cx->h_pin);
        cx->h_pin = nullptr; cx->h_cap = 0;
        RG_HIP(hipHostMalloc(&cx->h_pin, hb));
        cx->h_cap = hb;
    }
    {
        const size_t bytes = qn * 4;
        const int nth = bytes >= (2u << 20) ? 4 : 1;
        auto part = [&](int t) {
            const size_t lo = bytes * (size_t)t / (size_t)nth, hi = bytes * (size_t)(t + 1) / (size_t)nth;
            std::memcpy(static_cast<char *>(cx->h_pin) + lo, reinterpret_cast<const char *>(hq) + lo, hi - lo);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nth; ++t) th.emplace_back(part, t);
        part(0);
        for (auto &t : th) t.join();
    }
    RG_HIP(hipMemcpyAsync(cx->d_q, cx->h_pin, qn * 4, hipMemcpyHostToDevice, s));
    RG_HIP(hipMemsetAsync(cx->d_ids, 0, rn * 4, s));
    RG_HIP(hipMemsetAsync(cx->d_dist, 0, rn * 4, s));
    return search_dev(ix, cx, cx->d_q, n, d, k, L, cx->d_ids, cx->d_dist, cx->d_ch, cx->d_ch + n, s);
}

static rg_status host_search_end(HostSearch *hs, uint32_t k, uint32_t *out_ids, float *out_dists, uint32_t *out_cmps, uint32_t *out_hops) {
    if (!hs->cx) return RG_OK;
    rg_index *ix = hs->ix;
    SearchCtx *cx = hs->cx;
    (void)hipSetDevice(ix->device);
    rg_status st = finish_batches(ix, cx, cx->own, k);
    if (st == RG_OK || st == RG_ERR_NOT_ENOUGH) {
        const std::string msg = st != RG_OK ? rg_last_error() : "";
        const uint32_t n = hs->n;
        char *hp = static_cast<char *>(cx->h_pin);   // queries are on the device by now: the pinned buffer takes the results
        const size_t rb = (size_t)n * k * 4, cb = (size_t)n * 4;
        (void)hipMemcpyAsync(hp, cx->d_ids, rb, hipMemcpyDeviceToHost, cx->own);
        (void)hipMemcpyAsync(hp + rb, cx->d_dist, rb, hipMemcpyDeviceToHost, cx->own);
        (void)hipMemcpyAsync(hp + 2 * rb, cx->d_ch, 2 * cb, hipMemcpyDeviceToHost, cx->own);
        (void)hipStreamSynchronize(cx->own);
        std::memcpy(out_ids, hp, rb);
        std::memcpy(out_dists, hp + rb, rb);
        if (out_cmps) std::memcpy(out_cmps, hp + 2 * rb, cb);
        if (out_hops) std::memcpy(out_hops, hp + 2 * rb + cb, cb);
        if (st != RG_OK) set_error(st, msg);
    }
    release_ctx(ix, cx);
    hs->cx = nullptr;
    return st;
}

}  // namespace rg

using rg::set_error;

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
    rg::trace_stale("rg_index_close (entry: left by the searches)");
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    for (rg::SearchCtx *cx : ix->ctxs) rg::free_ctx(cx);
    if (ix->own_base) rg::dev_free(ix->d_base);
    if (ix->d_offsets) (void)hipFree(ix->d_offsets);
    if (ix->d_nbrs) (void)hipFree(ix->d_nbrs);
    rg::dev_free(ix->d_ell);
    if (ix->d_front_ids) (void)hipFree(ix->d_front_ids);
    rg::trace_stale("rg_index_close (after the frees)");
    rg::dev_free(ix->d_base_bf);
    rg::dev_free(ix->d_main);
    rg::dev_free(ix->d_etail);
    if (ix->d_tail_off) (void)hipFree(ix->d_tail_off);
    delete ix;
}

// adopt_base: the base is a dev_alloc'ed buffer the index takes over (rg_index_open_mem); otherwise it stays the caller's
static rg_status open_dev_impl(const float *d_base, uint32_t nd, uint32_t dim, uint32_t stride, const uint64_t *d_offsets,
                               const uint32_t *d_nbrs, uint32_t ep, int metric, int device, bool adopt_base, rg_index **out) {
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
    ix->own_base = adopt_base;
    if (const char *e = getenv("RG_FORCE_CSR")) ix->force_csr = atoi(e);
    st = rg::finish_graph(ix, d_offsets, d_nbrs);
    if (st != RG_OK) {
        if (adopt_base && ix->d_base == d_base) ix->own_base = false;     // the caller still owns what it passed in
        rg_index_close(ix);
        rg::dev_trim(device);
        return st;
    }
    rg::dev_trim(device);
    *out = ix;
    return RG_OK;
}

rg_status rg_index_open_dev(const float *d_base, uint32_t nd, uint32_t dim, uint32_t stride, const uint64_t *d_offsets,
                            const uint32_t *d_nbrs, uint32_t ep, int metric, int device, rg_index **out) {
    return open_dev_impl(d_base, nd, dim, stride, d_offsets, d_nbrs, ep, metric, device, false, out);
}

rg_status rg_index_open_mem(const float *base, uint32_t nd, uint32_t dim, uint32_t stride, const uint64_t *offsets,
                            const uint32_t *nbrs, uint32_t ep, int metric, int device, rg_index **out) {
    if (!out || !base || !offsets) return set_error(RG_ERR_ARG, "null argument");
    if (dim == 0 || stride < dim) return set_error(RG_ERR_ARG, "bad dim/stride");
    rg_status st = rg::pick_device(device);
    if (st != RG_OK) return st;
    // device copy at the aligned stride, zero padded (data_align, util.h:37-75); cosine rows are normalised first
    const uint32_t ad = rg::aligned_dim(dim);
    struct Big {      // (balanced over the memory classes when it is large: rg_mem.hip)
        float *p = nullptr;
        ~Big() { rg::dev_free(p); }
    } d_base;
    st = rg::dev_alloc_t(device, (size_t)nd * ad, &d_base.p);
    if (st != RG_OK) return st;
    const bool base_plain = rg::dev_last_plain();
    if (metric == RG_METRIC_COSINE || ad != dim || ad != stride) {
        std::vector<float> tmp((size_t)nd * ad, 0.0f);
        for (size_t i = 0; i < nd; ++i) std::memcpy(tmp.data() + i * ad, base + i * (size_t)stride, (size_t)dim * 4);
        if (metric == RG_METRIC_COSINE) rg_normalize_rows(tmp.data(), nd, ad, dim);
        st = rg::upload_staged(d_base.p, tmp.data(), tmp.size() * 4);
    } else {
        st = rg::upload_staged(d_base.p, base, (size_t)nd * ad * 4);
    }
    if (st != RG_OK) return st;
    const uint64_t ne = offsets[nd];
    rg::DevBuf<uint64_t> d_off;
    rg::DevBuf<uint32_t> d_nb;
    RG_HIP(d_off.alloc((size_t)nd + 1));
    RG_HIP(d_nb.alloc(ne));
    st = rg::upload_staged(d_off.p, offsets, ((size_t)nd + 1) * 8);
    if (st == RG_OK) st = rg::upload_staged(d_nb.p, nbrs, ne * 4);
    if (st != RG_OK) return st;
    rg_index *ix = nullptr;
    st = open_dev_impl(d_base.p, nd, ad, ad, d_off.p, d_nb.p, ep, metric, device, true, &ix);
    if (st != RG_OK) return st;
    d_base.p = nullptr;       // now owned by the index
    ix->n_plain_allocs += base_plain ? 1 : 0;
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

rg_status rg_index_open_multi(const char *base_fbin, const char *index_path, int metric, const int *devices, int ndev, rg_index **out) {
    if (!base_fbin || !index_path || !devices || !out || ndev <= 0) return set_error(RG_ERR_ARG, "null argument");
    for (int r = 0; r < ndev; ++r) out[r] = nullptr;
    uint32_t nb = 0, dim = 0, stride = 0, nd = 0, ep = 0;
    float *base = nullptr;
    uint64_t *off = nullptr;
    uint32_t *nbrs = nullptr;
    rg_status st = rg_fbin_load(base_fbin, &nb, &dim, &stride, &base);      // the files are read once, every replica is uploaded from memory
    if (st != RG_OK) return st;
    st = rg_graph_load(index_path, &nd, &ep, &off, &nbrs);
    if (st == RG_OK && nd != nb) st = set_error(RG_ERR_FORMAT, "index and base file disagree on the number of points");
    std::string msg = st != RG_OK ? rg_last_error() : "";
    for (int r = 0; r < ndev && st == RG_OK; ++r) {
        st = rg_index_open_mem(base, nb, dim, stride, off, nbrs, ep, metric, devices[r], &out[r]);
        if (st != RG_OK) msg = rg_last_error();
    }
    rg_free(base); rg_free(off); rg_free(nbrs);
    if (st != RG_OK) {
        for (int r = 0; r < ndev; ++r) { if (out[r]) rg_index_close(out[r]); out[r] = nullptr; }
        return set_error(st, msg);
    }
    return RG_OK;
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
    else if (!strcmp(name, "filter_fill")) ix->filter_fill = value;
    else if (!strcmp(name, "log_cap")) ix->log_cap_knob = value;
    else if (!strcmp(name, "log_budget_kb")) ix->log_budget_kb = value;
    else if (!strcmp(name, "visited_budget_kb")) ix->visited_budget_kb = value;
    else if (!strcmp(name, "visited_bytes")) ix->visited_bytes = value;
    else if (!strcmp(name, "gather_roll")) ix->gather_roll = value != 0;
    else if (!strcmp(name, "visited_uncached")) {
        if (value != ix->visited_uncached) {      // contexts re-allocate their words on the next exact-words launch
            std::lock_guard<std::mutex> lk(ix->mu);
            for (rg::SearchCtx *c : ix->ctxs) {
                rg::dev_free(c->d_visited);
                if (c->d_epoch) (void)hipFree(c->d_epoch);
                rg::dev_free(c->d_vtags);
                if (c->d_epoch8) (void)hipFree(c->d_epoch8);
                c->d_visited = c->d_epoch = c->d_vtags = c->d_epoch8 = nullptr;
                c->slots = c->tslots = 0;
            }
        }
        ix->visited_uncached = value;
    }
    else if (!strcmp(name, "query_in_lds")) ix->query_in_lds = value != 0;
    else if (!strcmp(name, "exact_filter")) ix->exact_filter = value != 0;
    else if (!strcmp(name, "lookahead")) ix->lookahead = value;
    else if (!strcmp(name, "gather_form")) ix->gather_form = value;
    else if (!strcmp(name, "count_in_k1")) ix->count_in_k1 = value;
    else if (!strcmp(name, "lset_bytes")) ix->lset_bytes = value;
    else if (!strcmp(name, "lset_tags")) ix->lset_tags = value;
    else if (!strcmp(name, "front_set")) ix->front_set = value;
    else if (!strcmp(name, "hub_bits")) ix->hub_bits = value;
    else if (!strcmp(name, "hub_pct")) ix->hub_pct = value;
    else if (!strcmp(name, "adaptive")) ix->adaptive = value != 0;
    else if (!strcmp(name, "lset")) ix->lset = value;
    else if (!strcmp(name, "log_early")) ix->log_early = value != 0;
    else if (!strcmp(name, "shared_frontier")) ix->shared_frontier = value != 0;
    else if (!strcmp(name, "filter_min_indeg")) ix->filter_min_indeg = value;
    else if (!strcmp(name, "multi_expand")) ix->multi_expand = value;
    else if (!strcmp(name, "split_rows")) ix->split_rows = value != 0;
    else if (!strcmp(name, "fast_bf16")) {
        // opt-in, NOT parity (see rg.h): the bf16 copy of the base is made on first use
        if (value && !ix->d_base_bf) {
            if (hipSetDevice(ix->device) != hipSuccess) return set_error(RG_ERR_DEVICE, "cannot select the index device");
            ix->stride_bf = (ix->dim + 127u) / 128u * 128u;
            {
                rg_status as = rg::dev_alloc_t(ix->device, (size_t)ix->nd * ix->stride_bf, &ix->d_base_bf);
                ix->n_plain_allocs += as == RG_OK && rg::dev_last_plain() ? 1 : 0;
                rg::dev_trim(ix->device);
                if (as != RG_OK) return as;
            }
            hipLaunchKernelGGL(rg::rg_base_to_bf16_kernel, dim3(ix->num_cu * 8), dim3(256), 0, 0, ix->d_base, ix->nd, ix->dim, ix->stride,
                               ix->d_base_bf, ix->stride_bf);
            RG_HIP(hipGetLastError());
            RG_HIP(hipDeviceSynchronize());
        }
        ix->fast_bf16 = value != 0;
    }
    else if (!strcmp(name, "count_table_log2")) { ix->count_table_log2 = value > 0 ? value : 15; ix->count_table_auto = value <= 0; }
    else if (!strcmp(name, "count_full_ids")) ix->count_full_ids = value != 0;
    else return set_error(RG_ERR_ARG, "unknown knob");
    return RG_OK;
}

rg_status rg_index_stat(const rg_index *ixc, const char *name, uint64_t *value) {
    if (!ixc || !name || !value) return set_error(RG_ERR_ARG, "null argument");
    rg_index *ix = const_cast<rg_index *>(ixc);
    std::lock_guard<std::mutex> lk(ix->mu);
    if (!strcmp(name, "batches_lset")) *value = ix->n_batches_lset;
    else if (!strcmp(name, "batches_filter_log")) *value = ix->n_batches_filter_log;
    else if (!strcmp(name, "batches_exact_hbm")) *value = ix->n_batches_exact_hbm;
    else if (!strcmp(name, "batches_filter_only")) *value = ix->n_batches_filter_only;
    else if (!strcmp(name, "lset_left")) *value = ix->n_lset_left;
    else if (!strcmp(name, "recounted")) *value = ix->n_recounted;
    else if (!strcmp(name, "hub_levels")) *value = ix->hub_levels ? 1 : 0;
    else if (!strcmp(name, "placement_balanced")) *value = ix->n_plain_allocs == 0 ? 1 : 0;
    else if (!strcmp(name, "plain_allocs")) *value = ix->n_plain_allocs;
    else if (!strcmp(name, "base_copied")) *value = ix->d_main ? 2 : (ix->base_copied ? 1 : 0);
    else if (!strcmp(name, "hub_m_last")) *value = ix->hub_m_last;
    else return set_error(RG_ERR_ARG, "unknown counter");
    return RG_OK;
}

rg_status rg_index_debug_ell(const rg_index *ix, uint32_t *host_out, uint64_t *nwords, uint32_t *stride) {
    if (!ix || !nwords || !stride) return set_error(RG_ERR_ARG, "null argument");
    if (!ix->d_ell) return set_error(RG_ERR_ARG, "the index keeps a CSR adjacency");
    *nwords = (uint64_t)ix->nd * ix->ell_stride;
    *stride = ix->ell_stride;
    if (host_out) {
        RG_HIP(hipSetDevice(ix->device));
        RG_HIP(hipMemcpy(host_out, ix->d_ell, (size_t)*nwords * 4, hipMemcpyDeviceToHost));
    }
    return RG_OK;
}

rg_status rg_projection_ep_dev(const float *d_base, uint32_t nd, uint32_t dim, uint32_t stride, int device, uint32_t *out_ep) {
    if (!d_base || !out_ep || nd == 0) return set_error(RG_ERR_ARG, "null argument");
    if (dim == 0 || dim % 4 || stride % 4 || stride < dim || ((uintptr_t)d_base & 15))
        return set_error(RG_ERR_ARG, "device base must be 16-byte aligned with dim % 4 == 0 and stride % 4 == 0");
    rg_status st = rg::pick_device(device);
    if (st != RG_OK) return st;
    rg::DevBuf<float> d_sum;
    rg::DevBuf<unsigned long long> d_best;
    RG_HIP(d_sum.alloc(dim));
    RG_HIP(d_best.alloc(1));
    hipLaunchKernelGGL(rg::rg_centroid_sum_kernel, dim3((dim + 31) / 32), dim3(256), 0, 0, d_base, nd, dim, stride, d_sum.p);
    std::vector<float> center(dim);
    RG_HIP(hipMemcpy(center.data(), d_sum.p, (size_t)dim * 4, hipMemcpyDeviceToHost));
    for (uint32_t d = 0; d < dim; ++d) center[d] /= (float)nd;        // :2014-2016, on the host: the same division as the host form
    RG_HIP(hipMemcpy(d_sum.p, center.data(), (size_t)dim * 4, hipMemcpyHostToDevice));
    RG_HIP(hipMemset(d_best.p, 0xff, 8));
    hipDeviceProp_t prop;
    RG_HIP(hipGetDeviceProperties(&prop, device));
    const uint32_t grid = std::min<uint32_t>((nd + 255) / 256, (uint32_t)prop.multiProcessorCount * 8u);
    hipLaunchKernelGGL(rg::rg_centroid_argmin_kernel, dim3(grid), dim3(256), (size_t)dim * 4, 0, d_base, nd, dim, stride, d_sum.p, d_best.p);
    unsigned long long best = 0;
    RG_HIP(hipMemcpy(&best, d_best.p, 8, hipMemcpyDeviceToHost));
    RG_HIP(hipGetLastError());
    *out_ep = (uint32_t)(best & 0xffffffffu);
    return RG_OK;
}

rg_status rg_search_dev(rg_index *ix, const float *d_queries, uint32_t nq, uint32_t qstride, uint32_t k, uint32_t L_pq,
                        uint32_t *d_ids, float *d_dists, uint32_t *d_cmps, uint32_t *d_hops, void *stream) {
    rg_status st = rg::check_search_args(ix, qstride, k, L_pq);
    if (st != RG_OK) return st;
    if (nq == 0) return RG_OK;
    RG_HIP(hipSetDevice(ix->device));
    rg::SearchCtx *cx = nullptr;
    st = rg::acquire_ctx(ix, (hipStream_t)stream, false, &cx);
    if (st != RG_OK) return st;
    st = rg::search_dev(ix, cx, d_queries, nq, qstride, k, L_pq, d_ids, d_dists, d_cmps, d_hops, (hipStream_t)stream);
    if (st != RG_OK) rg::release_ctx(ix, cx);
    return st;
}

rg_status rg_search_prepare(rg_index *ix, void *stream, uint32_t nq, uint32_t L_pq) {
    if (!ix) return set_error(RG_ERR_ARG, "null index");
    if (nq == 0) return RG_OK;
    RG_HIP(hipSetDevice(ix->device));
    rg::SearchCtx *cx = nullptr;
    rg_status st = rg::acquire_ctx(ix, (hipStream_t)stream, false, &cx);
    if (st != RG_OK) return st;
    if (ix->visited_mode == 2) st = rg::ensure_qlog(ix, cx, nq);
    // The exact set in HBM (one epoch byte per node and slot: nd bytes x the launch's grid, 19 GiB for a wide beam over 10M
    // nodes) is allocated ahead only where a launch will use it: visited = 0, or -- adaptive default -- a beam width from
    // which the exact set already won its timed trial.  A default-mode index that never leaves its filter + log form never
    // pays for the tags; the first trial batch of a width allocates them itself (and is not used as a measurement).
    uint32_t exact_from_L;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        exact_from_L = ix->exact_from_L;
    }
    if (st == RG_OK && (ix->visited_mode == 0 || (ix->visited_mode == 2 && L_pq >= exact_from_L))) {
        rg::K1Plan plan;      // the grid (= slots) and the tag form of exactly the launch rg_search_dev will make
        st = rg::plan_k1(ix, 0, nq, L_pq, false, false, false, (hipStream_t)stream, &plan);
        if (st == RG_OK) st = rg::ensure_visited(ix, cx, plan.c.grid, plan.vbytes, (hipStream_t)stream);
        if (st == RG_ERR_OOM && ix->visited_mode == 2) st = RG_OK;   // the default falls back to its filter + log form
    }
    const std::string msg = st != RG_OK ? rg_last_error() : "";
    rg::release_ctx(ix, cx);
    if (st != RG_OK) return set_error(st, msg);
    return RG_OK;
}

rg_status rg_search_reuse_stats(rg_index *ix, void *stream, uint64_t *evaluations, uint64_t *distinct_rows, uint32_t *d_row_counts) {
    if (!ix || !evaluations || !distinct_rows) return set_error(RG_ERR_ARG, "null argument");
    RG_HIP(hipSetDevice(ix->device));
    hipStream_t s = (hipStream_t)stream;
    // the context whose logs are read is held like a host-form search holds its own (reserved: no other caller is handed
    // it, so no search can re-allocate or overwrite the logs under the two kernels below)
    rg::SearchCtx *cx = nullptr;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        for (rg::SearchCtx *c : ix->ctxs)
            if (c->key == s && !c->reserved && c->pending.empty() && c->d_qlog && c->log_holds) { cx = c; break; }
        if (cx) { cx->reserved = true; cx->keyed = true; }
    }
    if (!cx) return set_error(RG_ERR_ARG, "no id logs for this stream: the last batch on it must have run in the default visited mode "
                                          "(filter + log), in one piece, and have been waited for");
    auto done = [&](rg_status st, const std::string &msg) {
        {
            std::lock_guard<std::mutex> lk(ix->mu);
            cx->reserved = false;
            if (cx->pending.empty()) cx->keyed = false;
        }
        return st == RG_OK ? RG_OK : set_error(st, msg);
    };
    const size_t words = ((size_t)ix->nd + 31) / 32;
    rg::DevBuf<uint32_t> bm;
    rg::DevBuf<unsigned long long> tot;
    if (bm.alloc(words) != hipSuccess || tot.alloc(2) != hipSuccess) { (void)hipGetLastError(); return done(RG_ERR_OOM, "no room for the row bitmap"); }
    hipError_t e = hipMemsetAsync(bm.p, 0, words * 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(tot.p, 0, 16, s);
    if (e == hipSuccess) {
        // d_row_counts, when given: nd counters (rg.h), one per base row, incremented per read
        hipLaunchKernelGGL(rg::rg_log_mark_kernel, dim3(std::min<uint32_t>(cx->log_holds, 4096u)), dim3(256), 0, s, cx->d_qlog, cx->logcap, cx->d_qlog_n,
                           cx->log_holds, bm.p, tot.p, d_row_counts);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rg::rg_bitmap_count_kernel, dim3(2048), dim3(256), 0, s, bm.p, words, tot.p + 1);
        e = hipGetLastError();
    }
    unsigned long long h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, tot.p, 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return done(RG_ERR_DEVICE, std::string("rg_search_reuse_stats: ") + hipGetErrorString(e));
    *evaluations = h[0];
    *distinct_rows = h[1];
    return done(RG_OK, "");
}

rg_status rg_search_wait(rg_index *ix, void *stream) {
    if (!ix) return set_error(RG_ERR_ARG, "null index");
    RG_HIP(hipSetDevice(ix->device));
    rg::SearchCtx *cx = nullptr;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        for (rg::SearchCtx *c : ix->ctxs)
            if (c->keyed && !c->reserved && c->key == (hipStream_t)stream) { cx = c; break; }
    }
    if (!cx) {   // nothing in flight on this stream
        RG_HIP(hipStreamSynchronize((hipStream_t)stream));
        return RG_OK;
    }
    rg_status st = rg::finish_batches(ix, cx, (hipStream_t)stream, 0);
    std::string msg = st != RG_OK ? rg_last_error() : "";
    if (cx->deferred != RG_OK) {   // a "not enough results" of batches collected early (more than 32 pending): the oldest first
        if (st == RG_OK || st == RG_ERR_NOT_ENOUGH) { st = cx->deferred; msg = cx->deferred_msg; }
        cx->deferred = RG_OK;
        cx->deferred_msg.clear();
    }
    rg::release_ctx(ix, cx);
    if (st != RG_OK) return set_error(st, msg);
    return RG_OK;
}

// queries -> the index stride, zero padded; cosine queries normalised (test_search_roargraph.cpp:167-172)
static void stage_queries(const rg_index *ix, const float *queries, uint32_t nq, uint32_t qstride, std::vector<float> &hq) {
    const uint32_t d = ix->dim, use = std::min(qstride, d);
    hq.assign((size_t)nq * d, 0.0f);
    for (size_t i = 0; i < nq; ++i) std::memcpy(hq.data() + i * d, queries + i * (size_t)qstride, (size_t)use * 4);
    if (ix->metric == RG_METRIC_COSINE) rg_normalize_rows(hq.data(), nq, d, d);
}

rg_status rg_search(rg_index *ix, const float *queries, uint32_t nq, uint32_t qstride, uint32_t k, uint32_t L_pq,
                    uint32_t *out_ids, float *out_dists, uint32_t *out_cmps, uint32_t *out_hops) {
    if (!ix || !queries || !out_ids || !out_dists) return set_error(RG_ERR_ARG, "null argument");
    if (k > L_pq) return set_error(RG_ERR_ARG, "L_pq must greater or equal than k");
    if (k == 0) return set_error(RG_ERR_ARG, "k and L_pq must be positive");
    if (nq == 0) return RG_OK;
    std::vector<float> hq;
    stage_queries(ix, queries, nq, qstride, hq);
    rg::HostSearch hs;
    rg_status st = rg::host_search_begin(ix, hq.data(), nq, ix->dim, k, L_pq, &hs);
    if (st != RG_OK) {
        const std::string msg = rg_last_error();
        if (hs.cx) { (void)hipStreamSynchronize(hs.cx->own); rg::release_ctx(ix, hs.cx); }
        return set_error(st, msg);
    }
    return rg::host_search_end(&hs, k, out_ids, out_dists, out_cmps, out_hops);
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
    if (k == 0) return set_error(RG_ERR_ARG, "k and L_pq must be positive");
    if (nq == 0) return RG_OK;
    const uint32_t d = replicas[0]->dim;
    std::vector<float> hq;
    stage_queries(replicas[0], queries, nq, qstride, hq);
    std::vector<rg::HostSearch> S((size_t)nrep);
    const uint32_t per = (nq + (uint32_t)nrep - 1) / (uint32_t)nrep;
    rg_status first = RG_OK;
    std::string first_msg;
    for (int r = 0; r < nrep && first == RG_OK; ++r) {
        rg::HostSearch &hs = S[(size_t)r];
        hs.lo = std::min(nq, (uint32_t)r * per);
        const uint32_t n = std::min(nq, hs.lo + per) - hs.lo;
        rg_status st = rg::host_search_begin(replicas[r], hq.data() + (size_t)hs.lo * d, n, d, k, L_pq, &hs);
        if (st != RG_OK) {
            first = st; first_msg = rg_last_error();
            if (hs.cx) { (void)hipStreamSynchronize(hs.cx->own); rg::release_ctx(replicas[r], hs.cx); hs.cx = nullptr; }
        }
    }
    // every launched slice is waited for, whatever happened to the others; the first failure (in query order) is reported.
    // A slice's "not enough results" names the query inside the slice: re-base it to the whole batch.
    for (int r = 0; r < nrep; ++r) {
        rg::HostSearch &hs = S[(size_t)r];
        if (!hs.cx) continue;
        rg_status w = rg::host_search_end(&hs, k, out_ids + (size_t)hs.lo * k, out_dists + (size_t)hs.lo * k,
                                          out_cmps ? out_cmps + hs.lo : nullptr, out_hops ? out_hops + hs.lo : nullptr);
        if (w != RG_OK && first == RG_OK) {
            first = w; first_msg = rg_last_error();
            unsigned got = 0, exp = 0, qq = 0;
            if (w == RG_ERR_NOT_ENOUGH && sscanf(first_msg.c_str(), "not enough results: %u, expected: %u (query %u)", &got, &exp, &qq) == 3) {
                char buf[160];
                snprintf(buf, sizeof buf, "not enough results: %u, expected: %u (query %u)", got, exp, qq + hs.lo);
                first_msg = buf;
            }
        }
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
