// rg_build_prune.hip -- PruneProjectionBaseSearchCandidates (src/index_bipartite.cpp:1846-1940) on the GPU: the occlusion
// pruning of a node's phase-3 expansion list, one wave per node.  Host counterpart: Builder::prune_search (rg_build.cpp),
// which this kernel reproduces bit for bit -- same (distance, id) order, same distance routine (rg_device.h: the
// reference's compare()), same greedy scan -- so the GPU-assisted build produces the lists the host pruning would
// (RG_BUILD_VERIFY=1 compares them node by node).
//
// Why here: on a 10M-row build the host spent 80 s in these scans (every distance between two base rows is a DRAM miss
// on the host), four times the GPU's time for the searches that produce the lists; the scan is the same gather-score
// pattern as K1, and the expansion lists are already in HBM.
//
// Per node (all state wave-private, in LDS):
//   keys   u64[1024]   (orderable distance bits << 32 | id), bitonic-sorted: the pool in the reference's order
//   rows   the chosen neighbours' base rows, staged 4 per pass in the layout gather_score reads (filled by LDS-DMA)
//   qv     a ring of D candidate rows filled by LDS-DMA: the rows of the next D candidates of the sorted pool are in
//          flight while one is scored (a random 800-B row is 2 us away, scoring a candidate takes a fraction of that);
//          every fill is issued unconditionally, so the wait in front of a slot is an exact vmcnt
// occluded(p) = some chosen r has compare(p, r) < dist(p) (or r == p): the chosen rows are scored four per pass against
// the candidate, stopping at the first pass that holds an occluder (the answer does not depend on which one is found).
// The reference's second sweep only matters for the entries in front of the first sweep's start (rg_build.cpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>

#include "rg.h"
#include "rg_device.h"
#include "rg_index_struct.h"
#include "rg_internal.h"

namespace rg {

struct PruneParams {
    const float *base;
    uint32_t stride, dim;
    const uint2 *exp;        // [n][cap] expanded (distance bits, id) in pop order
    uint32_t cap;
    const uint32_t *nexp;    // [n]
    const uint32_t *have;    // [n][hs]: word 0 = length of the node's projection list, then its ids
    uint32_t hs;
    uint32_t node0, n, M;
    uint32_t *out;           // [n][M + 1]: word 0 = length (0xffffffff: left to the host), then the pruned list
    uint32_t stage_floats;   // floats of one 4-row pass: ceil(dim / 64) * 256
    // phase 1 (round 3): the same greedy scan is the first sweep of PruneBiSearchBaseGetBase (:1612-1694) -- pivot = the
    // training query's nearest base point, pool = its other near neighbours scored against the pivot (rg_knn_score_kernel).
    // pivots[i] replaces node0 + i; `have` is null (nothing to skip); topup = the rule's last loop (:1683-1689): the list
    // is filled up to M with the not yet chosen pool entries in sorted order, occluded or not.  A pool that names an id
    // twice is left to the host (the rule keeps the first occurrence of every id).
    const uint32_t *pivots;
    uint32_t topup;
};

constexpr uint32_t kPruneKeys = 1024;
constexpr int kQChunksMax = 4;   // candidate row in registers while in flight: up to 4 x 64 lanes x 16 B = dim <= 1024

constexpr int kPruneRing = 4;     // candidate rows in flight

// QC: 1-KiB chunks (64 lanes x 16 B) of a candidate row: dim <= 256 QC
template <bool L2, int QC>
__global__ void __launch_bounds__(64) rg_prune_search_kernel(PruneParams P) {
    constexpr int D = kPruneRing;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int g = lane >> 4;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);                 // kPruneKeys
    uint32_t *res_id = reinterpret_cast<uint32_t *>(keys + kPruneKeys);                      // 64
    uint32_t *have_l = res_id + 64;                                                          // 64
    float *qv = reinterpret_cast<float *>(have_l + 64);                                      // D slots of QC * 256 floats
    float *rows = qv + D * QC * 256;                                                         // passes * stage_floats
    const uint32_t nq4 = P.dim / 4;                                                          // float4 pieces of a row

    for (uint32_t i = blockIdx.x; i < P.n; i += gridDim.x) {
        const uint32_t node = P.pivots ? P.pivots[i] : P.node0 + i;
        uint32_t *out = P.out + (size_t)i * (P.M + 1);
        const uint32_t ne = P.nexp ? P.nexp[i] : P.cap;
        const uint32_t nh = P.have ? P.have[(size_t)i * P.hs] : 0u;
        if (ne > P.cap || ne > kPruneKeys || (P.have && (nh + 1 > P.hs || nh > 64))) {   // the host prunes this one
            if (lane == 0) out[0] = 0xffffffffu;
            continue;
        }
        // ---- the pool: expansion list without the node itself (:1203), as sortable keys
        const uint2 *e = P.exp + (size_t)i * P.cap;
        uint32_t n = 0;
        for (uint32_t j0 = 0; j0 < ne; j0 += kWave) {
            const uint32_t j = j0 + lane;
            uint2 v = make_uint2(0u, 0u);
            bool keep = false;
            if (j < ne) { v = e[j]; keep = v.y != node; }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                uint32_t b = v.x;
                if (b == 0x80000000u) b = 0u;                                    // -0.0 == +0.0 in the reference's order
                const uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
                keys[n + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)ord << 32) | v.y;
            }
            n += __popcll(m);
        }
        if (P.have && lane < (int)nh) have_l[lane] = P.have[(size_t)i * P.hs + 1 + lane];
        uint32_t N = 64;
        while (N < n) N <<= 1;
        for (uint32_t j = n + lane; j < N; j += kWave) keys[j] = ~0ull;
        lds_sync();
        // ---- std::sort by (distance, id) (:1863): bitonic network over N keys
        for (uint32_t k = 2; k <= N; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = lane; t < N / 2; t += kWave) {
                    const uint32_t lo = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), hi = lo | j;
                    const unsigned long long a = keys[lo], b = keys[hi];
                    const bool up = (lo & k) == 0u;
                    if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
                }
                lds_sync();
            }
        if (P.topup) {   // the same id twice (same distance: adjacent after the sort): the host's rule drops repeats first
            bool twice = false;
            for (uint32_t j = lane; j + 1 < n; j += kWave) twice = twice || keys[j] == keys[j + 1];
            if (__ballot(twice)) {
                if (lane == 0) out[0] = 0xffffffffu;
                lds_sync();
                continue;
            }
        }
        auto pool_id = [&](uint32_t j) { return (uint32_t)(keys[j] & 0xffffffffull); };
        auto pool_dist = [&](uint32_t j) {
            const uint32_t ord = (uint32_t)(keys[j] >> 32);
            return __uint_as_float((ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord);
        };
        // ---- first entry the projection list does not hold already (:1866-1871)
        uint32_t first = n;
        for (uint32_t j0 = 0; j0 < n && first == n; j0 += kWave) {
            const uint32_t j = j0 + lane;
            bool fresh = false;
            if (j < n) {
                const uint32_t id = pool_id(j);
                fresh = true;
                for (uint32_t x = 0; x < nh; ++x) fresh = fresh && have_l[x] != id;
            }
            const unsigned long long m = __ballot(fresh);
            if (m) first = j0 + (uint32_t)__builtin_ctzll(m);
        }
        uint32_t cnt = 0;
        // chosen neighbour: id + its base row into the staged passes (LDS-DMA; the scoring layout of rg_device.h)
        auto append = [&](uint32_t id) {
            if (lane == 0) res_id[cnt] = id;
            gather_issue(P.base + (size_t)id * P.stride, P.dim, g == (int)(cnt & 3u), rows + (size_t)(cnt >> 2) * P.stage_floats, lane);
            gather_wait(0);
            lds_sync();
            ++cnt;
        };
        // occluded(p, result): compare() of p against the chosen rows, four per pass
        auto occluded = [&](const float *q, uint32_t pid, float pd) -> bool {
            for (uint32_t ps = 0; 4u * ps < cnt; ++ps) {
                const float d = gather_score<L2>(rows + (size_t)ps * P.stage_floats, q, P.dim, lane);
                const uint32_t c = 4u * ps + (uint32_t)g;
                const bool hit = c < cnt && (res_id[c] == pid || d < pd);
                if (__ballot(hit)) return true;
            }
            return false;
        };
        // candidate rows in flight: slot s of the ring holds the candidate that is D ahead of the one last taken from it.
        // Every fill is issued unconditionally (lanes beyond the row re-read its start into the slot's padding, indices
        // beyond the pool re-read its last entry): QC loads per fill, so "at most (D - 1) QC outstanding" means slot s is in
        uint32_t xoff[QC];
#pragma unroll
        for (int c = 0; c < QC; ++c) { const uint32_t x = (uint32_t)(c * kWave + lane); xoff[c] = 4u * (x < nq4 ? x : 0u); }
#define RG_FETCH(s_, id_)                                                                  \
    {                                                                                      \
        const float *src_ = P.base + (size_t)(id_) * P.stride;                             \
        _Pragma("unroll") for (int c_ = 0; c_ < QC; ++c_)                                  \
            __builtin_amdgcn_global_load_lds((glb_ptr_t *)(src_ + xoff[c_]), (lds_ptr_t *)(qv + ((s_) * QC + c_) * 256), 16, 0, 0); \
    }
#define RG_ARRIVED(n_) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n_) : "memory"); __builtin_amdgcn_wave_barrier(); }
        if (first < n) {
            append(pool_id(first));
            // ---- first sweep (:1874-1907)
            const uint32_t j0 = first + 1;
#pragma unroll
            for (int s = 0; s < D; ++s) RG_FETCH(s, pool_id(min(j0 + (uint32_t)s, n - 1u)));
            bool more = j0 < n;
            for (uint32_t jb = j0; more; jb += D) {
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    const uint32_t j = jb + (uint32_t)s;
                    if (more && (j >= n || cnt >= P.M)) more = false;
                    if (more) {
                        RG_ARRIVED((D - 1) * QC);
                        const uint32_t pid = pool_id(j);
                        const float pd = pool_dist(j);
                        const bool occ = occluded(qv + s * QC * 256, pid, pd);
                        lds_sync();                                        // the slot has been read: refill it
                        RG_FETCH(s, pool_id(min(j + (uint32_t)D, n - 1u)));
                        if (!occ && pid != node) append(pid);
                    }
                }
            }
            // ---- second sweep (:1912-1926): only the entries in front of `first` are still undecided
            for (uint32_t j2 = 1; j2 < first && cnt < P.M; ++j2) {
                const uint32_t pid = pool_id(j2);
                const float pd = pool_dist(j2);
                lds_sync();
                RG_FETCH(0, pid);
                RG_ARRIVED(0);
                if (!occluded(qv, pid, pd) && pid != node) append(pid);
            }
        }
#undef RG_FETCH
#undef RG_ARRIVED
        lds_sync();
        if (P.topup)     // :1683-1689: the sorted pool from its second entry on, whatever was not chosen, until the list is full
            for (uint32_t j = 1; j < n && cnt < P.M; ++j) {
                const uint32_t pid = pool_id(j);
                const bool in = (uint32_t)lane < cnt && res_id[lane] == pid;
                if (!__ballot(in) && pid != node) {
                    if (lane == 0) res_id[cnt] = pid;
                    ++cnt;
                    lds_sync();
                }
            }
        if (lane == 0) out[0] = cnt;
        if ((uint32_t)lane < cnt) out[1 + lane] = res_id[lane];
        lds_sync();
    }
}

// Phase 1's pools: exp[i][c] = (bits of compare(base[knn[i][c]], base[knn[i][0]]), knn[i][c]) for c < ncol -- the near
// neighbours of training query i scored against its nearest base point (:1074-1082), with the exact routine of K1b.  One
// wave per query at a time, the pivot's row staged in LDS as the "query", the rows through a ring of R passes of four.
template <bool L2, int R>
__global__ void __launch_bounds__(64) rg_knn_score_kernel(const float *__restrict__ base, uint32_t stride, uint32_t dim,
                                                          const uint32_t *__restrict__ knn, uint32_t n, uint32_t kdim, uint32_t ncol,
                                                          uint2 *__restrict__ exp, uint32_t cap, uint32_t *__restrict__ pivots,
                                                          uint32_t stage_floats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x, g = lane >> 4;
    float *stage = reinterpret_cast<float *>(smem);
    float *qv = stage + (size_t)R * stage_floats;
    const uint32_t npass = (ncol + 3u) >> 2, lpp = loads_per_pass(dim);
    for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) {
        const uint32_t *ids = knn + (size_t)q * kdim;
        const uint32_t tgt = ids[0];
        if (lane == 0) pivots[q] = tgt;
        for (uint32_t i = lane; i < dim; i += kWave) qv[i] = base[(size_t)tgt * stride + i];
        wave_sync();
        auto issue = [&](uint32_t p, float *buf) {
            const uint32_t c = 4 * p + g;
            const bool act = c < ncol;
            gather_issue(base + (size_t)(act ? ids[c] : 0u) * stride, dim, act, buf, lane);
        };
        for (uint32_t p = 0; p < (uint32_t)R && p < npass; ++p) issue(p, stage + (size_t)p * stage_floats);
        for (uint32_t p = 0; p < npass; ++p) {
            const uint32_t last = min(npass, p + (uint32_t)R) - 1u;
            gather_wait((last - p) * lpp);
            float *buf = stage + (size_t)(p & (R - 1)) * stage_floats;
            const uint32_t c = 4 * p + g;
            const float d = gather_score<L2>(buf, qv, dim, lane);
            if (c < ncol && (lane & 15) == 0) exp[(size_t)q * cap + c] = make_uint2(__float_as_uint(d), ids[c]);
            lds_sync();
            if (p + R < npass) issue(p + R, buf);
        }
        wave_sync();
    }
}

// LDS bytes of one workgroup, 0 if the shape does not fit (then the host prunes)
static size_t prune_lds_bytes(uint32_t dim, uint32_t M, size_t lds_per_cu) {
    if (dim % 8 || dim > (uint32_t)kQChunksMax * kWave * 4 || M > 64) return 0;
    const size_t stage = (size_t)((dim + 63) / 64) * 256;
    const size_t qc = dim <= 256 ? 1 : dim <= 512 ? 2 : 4;   // 1-KiB chunks of a candidate slot
    const size_t b = (size_t)kPruneKeys * 8 + 64 * 4 + 64 * 4 + (size_t)kPruneRing * qc * 1024 + (size_t)((M + 3) / 4) * stage * 4;
    return b <= lds_per_cu ? (b + 15) / 16 * 16 : 0;
}

bool build_prune_supported(const rg_index *ix, uint32_t M, uint32_t exp_cap) {
    return exp_cap <= kPruneKeys && prune_lds_bytes(ix->dim, M, ix->lds_per_cu) != 0;   // longer lists: the host prunes
}

rg_status build_prune_dev(rg_index *ix, uint32_t node0, uint32_t n, uint32_t M, const uint2_pod *d_exp, uint32_t exp_cap,
                          const uint32_t *d_nexp, const uint32_t *d_have, uint32_t hs, uint32_t *d_out, void *stream) {
    if (!ix || !d_exp || !d_nexp || !d_have || !d_out) return set_error(RG_ERR_ARG, "null argument");
    if (n == 0) return RG_OK;
    const size_t lds = prune_lds_bytes(ix->dim, M, ix->lds_per_cu);
    if (!lds) return set_error(RG_ERR_ARG, "pruning kernel: shape not supported");
    if (hipSetDevice(ix->device) != hipSuccess) return set_error(RG_ERR_DEVICE, "cannot select the index device");
    PruneParams P;
    P.base = ix->d_base; P.stride = ix->stride; P.dim = ix->dim;
    P.exp = reinterpret_cast<const uint2 *>(d_exp); P.cap = exp_cap; P.nexp = d_nexp;
    P.have = d_have; P.hs = hs; P.node0 = node0; P.n = n; P.M = M; P.out = d_out;
    P.pivots = nullptr; P.topup = 0;
    P.stage_floats = (uint32_t)((ix->dim + 63) / 64) * 256u;
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, ix->lds_per_cu / lds));
    const dim3 grid(std::min<uint32_t>(n, (uint32_t)ix->num_cu * per_cu));
    const bool l2 = ix->metric == RG_METRIC_L2;
    const int qc = ix->dim <= 256 ? 1 : ix->dim <= 512 ? 2 : 4;
#define RG_PRUNE_LAUNCH(L2_, QC_)                                                                                              \
    do {                                                                                                                       \
        auto kern = rg_prune_search_kernel<L2_, QC_>;                                                                          \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return set_error(RG_ERR_DEVICE, "pruning kernel: cannot reserve its LDS");                                         \
        hipLaunchKernelGGL(kern, grid, dim3(kWave), lds, (hipStream_t)stream, P);                                              \
    } while (0)
    if (l2) { if (qc == 1) RG_PRUNE_LAUNCH(true, 1); else if (qc == 2) RG_PRUNE_LAUNCH(true, 2); else RG_PRUNE_LAUNCH(true, 4); }
    else { if (qc == 1) RG_PRUNE_LAUNCH(false, 1); else if (qc == 2) RG_PRUNE_LAUNCH(false, 2); else RG_PRUNE_LAUNCH(false, 4); }
#undef RG_PRUNE_LAUNCH
    if (hipGetLastError() != hipSuccess) return set_error(RG_ERR_DEVICE, "pruning kernel launch failed");
    return RG_OK;
}

// Phase 1 on the GPU: the pruned list of every training query of a chunk (PruneBiSearchBaseGetBase over its knn row).
// d_knn [n][kdim] (device), ncol = min(kdim, M_sq) columns are used; d_exp [n][cap] and d_pivots [n] are scratch; d_out
// [n][M + 1] as build_prune_dev writes it.
bool build_prune_knn_supported(uint32_t dim, uint32_t M, uint32_t ncol, size_t lds_per_cu) {
    return ncol >= 1 && ncol <= kPruneKeys && prune_lds_bytes(dim, M, lds_per_cu) != 0;
}

rg_status build_prune_knn_dev(const float *d_base, uint32_t dim, uint32_t stride, int metric, int device, int num_cu, size_t lds_per_cu,
                              const uint32_t *d_knn, uint32_t n, uint32_t kdim, uint32_t ncol, uint32_t M, uint2_pod *d_exp, uint32_t cap,
                              uint32_t *d_pivots, uint32_t *d_out, void *stream) {
    if (!d_base || !d_knn || !d_exp || !d_pivots || !d_out) return set_error(RG_ERR_ARG, "null argument");
    if (n == 0) return RG_OK;
    const size_t lds = prune_lds_bytes(dim, M, lds_per_cu);
    if (!lds || ncol > cap || cap > kPruneKeys) return set_error(RG_ERR_ARG, "pruning kernel: shape not supported");
    if (hipSetDevice(device) != hipSuccess) return set_error(RG_ERR_DEVICE, "cannot select the build device");
    const bool l2 = metric == RG_METRIC_L2;
    {
        constexpr int R = 4;
        const uint32_t stage_floats = ((dim + 63) / 64) * 256;
        const size_t slds = (size_t)R * stage_floats * 4 + (size_t)dim * 4;
        const dim3 grid(std::min<uint32_t>(n, (uint32_t)num_cu * 16u));
        if (l2) hipLaunchKernelGGL((rg_knn_score_kernel<true, R>), grid, dim3(kWave), slds, (hipStream_t)stream, d_base, stride, dim, d_knn, n, kdim, ncol,
                                   reinterpret_cast<uint2 *>(d_exp), cap, d_pivots, stage_floats);
        else hipLaunchKernelGGL((rg_knn_score_kernel<false, R>), grid, dim3(kWave), slds, (hipStream_t)stream, d_base, stride, dim, d_knn, n, kdim, ncol,
                                reinterpret_cast<uint2 *>(d_exp), cap, d_pivots, stage_floats);
    }
    PruneParams P;
    P.base = d_base; P.stride = stride; P.dim = dim;
    P.exp = reinterpret_cast<const uint2 *>(d_exp); P.cap = cap; P.nexp = nullptr;
    P.have = nullptr; P.hs = 0; P.node0 = 0; P.n = n; P.M = M; P.out = d_out;
    P.pivots = d_pivots; P.topup = 1;
    P.stage_floats = (uint32_t)((dim + 63) / 64) * 256u;
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, lds_per_cu / lds));
    const dim3 grid(std::min<uint32_t>(n, (uint32_t)num_cu * per_cu));
    const int qc = dim <= 256 ? 1 : dim <= 512 ? 2 : 4;
#define RG_PRUNE_LAUNCH(L2_, QC_)                                                                                              \
    do {                                                                                                                       \
        auto kern = rg_prune_search_kernel<L2_, QC_>;                                                                          \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return set_error(RG_ERR_DEVICE, "pruning kernel: cannot reserve its LDS");                                         \
        hipLaunchKernelGGL(kern, grid, dim3(kWave), lds, (hipStream_t)stream, P);                                              \
    } while (0)
    if (l2) { if (qc == 1) RG_PRUNE_LAUNCH(true, 1); else if (qc == 2) RG_PRUNE_LAUNCH(true, 2); else RG_PRUNE_LAUNCH(true, 4); }
    else { if (qc == 1) RG_PRUNE_LAUNCH(false, 1); else if (qc == 2) RG_PRUNE_LAUNCH(false, 2); else RG_PRUNE_LAUNCH(false, 4); }
#undef RG_PRUNE_LAUNCH
    if (hipGetLastError() != hipSuccess) return set_error(RG_ERR_DEVICE, "phase-1 pruning launch failed");
    return RG_OK;
}

}  // namespace rg
