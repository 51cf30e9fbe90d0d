// K4 experiment harness (not part of the product): where does the exact-distinct-count pass spend its time?
// Synthetic logs: nq queries x n ids (uniform random in [0, nd), a few duplicates).  Variants:
//   0 = product form (atomicCAS insert loop)      1 = plain read / conditional write (racy: timing only)
//   2 = loads + table clear only                  3 = phased non-atomic protocol (read-walk / barrier / write / barrier)
// build: hipcc --offload-arch=gfx950 -O3 -o k4mb scripts/exp/k4_microbench.hip ; run: ./k4mb [n] [nq]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#include <set>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V>
__global__ void __launch_bounds__(1024) k4(const uint32_t *qlog, uint32_t logcap, const uint32_t *qlog_n, uint32_t nq,
                                          uint32_t *out, uint32_t tbits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    __shared__ uint32_t s_cnt;
    const uint32_t T = 1u << tbits, cap = (T / 4u) * 3u;
    const int tid = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const uint32_t n = qlog_n[q];
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const uint32_t *log = qlog + (size_t)q * logcap;
        const uint32_t parts = (n + cap - 1) / cap;
        uint32_t mine = 0;
        for (uint32_t p = 0; p < parts; ++p) {
            for (uint32_t i = tid; i < T; i += blockDim.x) tab[i] = 0xffffffffu;
            __syncthreads();
            for (uint32_t i0 = tid; i0 < n; i0 += blockDim.x * 8u) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t i = i0 + (uint32_t)u * blockDim.x;
                    v[u] = i < n ? log[i] : 0xffffffffu;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t id = v[u];
                    if (id == 0xffffffffu) continue;
                    if (parts > 1 && ((id * 0x85EBCA6Bu) >> 16) % parts != p) continue;
                    uint32_t slot = (id * 0x9E3779B1u) >> (32u - tbits);
                    if (V == 2) { mine += slot & 1; continue; }
                    for (;;) {
                        uint32_t old;
                        if (V == 0) old = atomicCAS(&tab[slot], 0xffffffffu, id);
                        else { old = tab[slot]; if (old == 0xffffffffu) tab[slot] = id; }
                        if (old == 0xffffffffu) { ++mine; break; }
                        if (old == id) break;
                        slot = (slot + 1u) & (T - 1u);
                    }
                }
            }
            __syncthreads();
        }
        for (int o = 32; o; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o, 64);
        if ((tid & 63) == 0) atomicAdd(&s_cnt, mine);
        __syncthreads();
        if (tid == 0) out[q] = s_cnt;
        __syncthreads();
    }
}

// variant 3: phased protocol.  CH ids per thread per chunk held in registers; a round = WALK parallel read steps,
// barrier(any pending?), writes into slots seen empty, barrier.  Occupied slots never change, so reads are safe at any
// time; a slot is written only in the write phase of the round in which it was seen empty, and every writer re-reads it
// after the closing barrier (the winner sees its own id, losers move on).  Count = occupied slots at the end.
template <int CH, int WALK>
__global__ void __launch_bounds__(1024) k4_phased(const uint32_t *qlog, uint32_t logcap, const uint32_t *qlog_n, uint32_t nq,
                                                 uint32_t *out, uint32_t tbits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    __shared__ uint32_t s_cnt;
    const uint32_t T = 1u << tbits, cap = (T / 4u) * 3u, mask = T - 1u;
    const int tid = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const uint32_t n = qlog_n[q];
        if (tid == 0) s_cnt = 0;
        const uint32_t *log = qlog + (size_t)q * logcap;
        const uint32_t parts = (n + cap - 1) / cap;
        uint32_t total = 0;
        for (uint32_t p = 0; p < parts; ++p) {
            for (uint32_t i = tid * 4; i < T; i += blockDim.x * 4) *reinterpret_cast<uint4 *>(tab + i) = make_uint4(~0u, ~0u, ~0u, ~0u);
            __syncthreads();
            for (uint32_t c0 = 0; c0 < n; c0 += blockDim.x * CH) {
                uint32_t id[CH], slot[CH];
                uint32_t pend = 0;   // bit u: id[u] still to be placed
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const uint32_t i = c0 + (uint32_t)u * blockDim.x + tid;
                    id[u] = i < n ? log[i] : 0xffffffffu;
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    bool act = id[u] != 0xffffffffu;
                    if (parts > 1 && ((id[u] * 0x85EBCA6Bu) >> 16) % parts != p) act = false;
                    slot[u] = (id[u] * 0x9E3779B1u) >> (32u - tbits);
                    pend |= act ? (1u << u) : 0u;
                }
                for (;;) {
                    uint32_t wr = 0;
#pragma unroll
                    for (int w = 0; w < WALK; ++w) {
                        uint32_t v[CH];
#pragma unroll
                        for (int u = 0; u < CH; ++u) v[u] = tab[slot[u]];
                        wr = 0;
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            if (!(pend >> u & 1u)) continue;
                            if (v[u] == id[u]) pend &= ~(1u << u);
                            else if (v[u] == 0xffffffffu) wr |= 1u << u;
                            else slot[u] = (slot[u] + 1u) & mask;
                        }
                    }
                    if (!__syncthreads_or(pend != 0)) break;
#pragma unroll
                    for (int u = 0; u < CH; ++u)
                        if (wr >> u & 1u) tab[slot[u]] = id[u];
                    __syncthreads();
                }
            }
            uint32_t occ = 0;
            for (uint32_t i = tid * 4; i < T; i += blockDim.x * 4) {
                const uint4 t = *reinterpret_cast<const uint4 *>(tab + i);
                occ += (t.x != ~0u) + (t.y != ~0u) + (t.z != ~0u) + (t.w != ~0u);
            }
            total += occ;
            __syncthreads();
        }
        for (int o = 32; o; o >>= 1) total += (uint32_t)__shfl_xor((int)total, o, 64);
        if ((tid & 63) == 0) atomicAdd(&s_cnt, total);
        __syncthreads();
        if (tid == 0) out[q] = s_cnt;
        __syncthreads();
    }
}

// variant 4: 4-slot buckets read with one ds_read_b128, double hashing between buckets, CAS only on the chosen empty
// slot (linear probing at load 0.67 has probe tails of dozens of slots, and a wave pays the longest of its 64 lanes)
template <int UN>
__global__ void __launch_bounds__(1024) k4_bucket(const uint32_t *qlog, uint32_t logcap, const uint32_t *qlog_n, uint32_t nq,
                                                 uint32_t *out, uint32_t tbits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    __shared__ uint32_t s_cnt;
    const uint32_t T = 1u << tbits, cap = (T / 4u) * 3u, bbits = tbits - 2u, bmask = (1u << bbits) - 1u;
    const int tid = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const uint32_t n = qlog_n[q];
        if (tid == 0) s_cnt = 0;
        const uint32_t *log = qlog + (size_t)q * logcap;
        const uint32_t parts = (n + cap - 1) / cap;
        uint32_t mine = 0;
        for (uint32_t p = 0; p < parts; ++p) {
            for (uint32_t i = tid * 4; i < T; i += blockDim.x * 4) *reinterpret_cast<uint4 *>(tab + i) = make_uint4(~0u, ~0u, ~0u, ~0u);
            __syncthreads();
            for (uint32_t i0 = tid; i0 < n; i0 += blockDim.x * UN) {
                uint32_t v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const uint32_t i = i0 + (uint32_t)u * blockDim.x;
                    v[u] = i < n ? log[i] : 0xffffffffu;
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const uint32_t id = v[u];
                    if (id == 0xffffffffu) continue;
                    const uint32_t h = id * 0x9E3779B1u;
                    if (parts > 1 && ((id * 0x85EBCA6Bu) >> 16) % parts != p) continue;
                    uint32_t b = h >> (32u - bbits);
                    const uint32_t step = ((h >> 3) | 1u) & bmask;
                    for (;;) {
                        const uint4 t = *reinterpret_cast<const uint4 *>(tab + 4u * b);
                        if (t.x == id || t.y == id || t.z == id || t.w == id) break;
                        const int e = t.x == ~0u ? 0 : t.y == ~0u ? 1 : t.z == ~0u ? 2 : t.w == ~0u ? 3 : 4;
                        if (e == 4) { b = (b + step) & bmask; continue; }
                        const uint32_t old = atomicCAS(&tab[4u * b + e], 0xffffffffu, id);
                        if (old == 0xffffffffu) { ++mine; break; }
                        if (old == id) break;
                    }
                }
            }
            __syncthreads();
        }
        for (int o = 32; o; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o, 64);
        if ((tid & 63) == 0) atomicAdd(&s_cnt, mine);
        __syncthreads();
        if (tid == 0) out[q] = s_cnt;
        __syncthreads();
    }
}

// variant 5: variant 4 with G ids per thread advanced in lock-step (G bucket reads in flight, then G CAS in flight):
// the insert chain is latency bound, so each wave carries G independent chains
template <int G>
__global__ void __launch_bounds__(1024) k4_bucket_ilp(const uint32_t *qlog, uint32_t logcap, const uint32_t *qlog_n, uint32_t nq,
                                                      uint32_t *out, uint32_t tbits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    __shared__ uint32_t s_cnt;
    const uint32_t T = 1u << tbits, cap = (T / 4u) * 3u, bbits = tbits - 2u, bmask = (1u << bbits) - 1u;
    const int tid = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const uint32_t n = qlog_n[q];
        if (tid == 0) s_cnt = 0;
        const uint32_t *log = qlog + (size_t)q * logcap;
        const uint32_t parts = (n + cap - 1) / cap;
        uint32_t mine = 0;
        for (uint32_t p = 0; p < parts; ++p) {
            for (uint32_t i = tid * 4; i < T; i += blockDim.x * 4) *reinterpret_cast<uint4 *>(tab + i) = make_uint4(~0u, ~0u, ~0u, ~0u);
            __syncthreads();
            for (uint32_t i0 = tid; i0 < n; i0 += blockDim.x * G) {
                uint32_t id[G], b[G], step[G];
                uint32_t pend = 0;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t i = i0 + (uint32_t)g * blockDim.x;
                    id[g] = i < n ? log[i] : 0xffffffffu;
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t h = id[g] * 0x9E3779B1u;
                    bool act = id[g] != 0xffffffffu;
                    if (parts > 1 && ((id[g] * 0x85EBCA6Bu) >> 16) % parts != p) act = false;
                    b[g] = h >> (32u - bbits);
                    step[g] = ((h >> 3) | 1u) & bmask;
                    pend |= act ? 1u << g : 0u;
                }
                while (pend) {
                    uint4 t[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) t[g] = *reinterpret_cast<const uint4 *>(tab + 4u * b[g]);
                    uint32_t old[G]; int e[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        e[g] = -1;
                        if (!(pend >> g & 1u)) continue;
                        if (t[g].x == id[g] || t[g].y == id[g] || t[g].z == id[g] || t[g].w == id[g]) { pend &= ~(1u << g); continue; }
                        e[g] = t[g].x == ~0u ? 0 : t[g].y == ~0u ? 1 : t[g].z == ~0u ? 2 : t[g].w == ~0u ? 3 : 4;
                        if (e[g] == 4) { b[g] = (b[g] + step[g]) & bmask; e[g] = -1; }
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        if (e[g] >= 0) old[g] = atomicCAS(&tab[4u * b[g] + e[g]], 0xffffffffu, id[g]);
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        if (e[g] >= 0) {
                            if (old[g] == 0xffffffffu) { ++mine; pend &= ~(1u << g); }
                            else if (old[g] == id[g]) pend &= ~(1u << g);
                        }
                }
            }
            __syncthreads();
        }
        for (int o = 32; o; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o, 64);
        if ((tid & 63) == 0) atomicAdd(&s_cnt, mine);
        __syncthreads();
        if (tid == 0) out[q] = s_cnt;
        __syncthreads();
    }
}

// variant 6: 16-bit remainders of a bijective hash (id * odd mod 2^id_bits): bucket = top 13 bits (8192 buckets of
// 8 halves = one ds_read_b128), remainder = the rest (<= 15 bits, 0xffff = empty).  An id lives only in its home
// bucket; ids whose bucket is full go to a small exact overflow table of full ids.  65536 slots in 128 KiB: one pass
// up to ~45k ids.  Dynamic query scheduling (atomic counter).
__device__ inline uint32_t half_of(const uint4 &t, int e) {
    const uint32_t w = e < 2 ? t.x : e < 4 ? t.y : e < 6 ? t.z : t.w;
    return (e & 1) ? w >> 16 : w & 0xffffu;
}
__global__ void __launch_bounds__(1024) k4_half(const uint32_t *qlog, uint32_t logcap, const uint32_t *qlog_n, uint32_t nq,
                                                uint32_t *out, uint32_t id_bits, uint32_t *work) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);            // 8192 buckets x 4 words
    uint32_t *ovf = tab + 32768;                                    // 4096 full ids
    __shared__ uint32_t s_cnt, s_q, s_fail;
    const uint32_t OV = 4096u, rbits = id_bits - 13u, hmask = id_bits >= 32 ? ~0u : (1u << id_bits) - 1u;
    const int tid = threadIdx.x;
    for (;;) {
        if (tid == 0) { s_q = atomicAdd(work, 1u); s_cnt = 0; s_fail = 0; }
        for (uint32_t i = tid * 4; i < 32768u + OV; i += blockDim.x * 4) *reinterpret_cast<uint4 *>(tab + i) = make_uint4(~0u, ~0u, ~0u, ~0u);
        __syncthreads();
        const uint32_t q = s_q;
        if (q >= nq) break;
        const uint32_t n = qlog_n[q];
        const uint32_t *log = qlog + (size_t)q * logcap;
        uint32_t mine = 0;
        uint32_t nx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = tid + (uint32_t)u * blockDim.x;
            nx[u] = i < n ? log[i] : 0xffffffffu;
        }
        for (uint32_t i0 = tid; i0 < n; i0 += blockDim.x * 4u) {
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = nx[u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // next round's ids are in flight while this round inserts
                const uint32_t i = i0 + blockDim.x * 4u + (uint32_t)u * blockDim.x;
                nx[u] = i < n ? log[i] : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t id = v[u];
                if (id == 0xffffffffu) continue;
                const uint32_t h = (id * 0x9E3779B1u) & hmask;
                const uint32_t b = h >> rbits, rem = h & ((1u << rbits) - 1u);
                for (;;) {
                    const uint4 t = *reinterpret_cast<const uint4 *>(tab + 4u * b);
                    int e = 8; bool found = false;
#pragma unroll
                    for (int k = 7; k >= 0; --k) {
                        const uint32_t hv = half_of(t, k);
                        found |= hv == rem;
                        if (hv == 0xffffu) e = k;
                    }
                    if (found) break;
                    if (e == 8) {   // home bucket full: exact overflow table
                        uint32_t slot = (id * 0x85EBCA6Bu) >> 20;
                        uint32_t probes = 0;
                        for (;;) {
                            const uint32_t old = atomicCAS(&ovf[slot], 0xffffffffu, id);
                            if (old == 0xffffffffu) { ++mine; break; }
                            if (old == id) break;
                            slot = (slot + 1u) & (OV - 1u);
                            if (++probes >= OV) { s_fail = 1; break; }
                        }
                        break;
                    }
                    const uint32_t w = e < 2 ? t.x : e < 4 ? t.y : e < 6 ? t.z : t.w;
                    const uint32_t nw = (e & 1) ? (w & 0x0000ffffu) | (rem << 16) : (w & 0xffff0000u) | rem;
                    const uint32_t old = atomicCAS(&tab[4u * b + (e >> 1)], w, nw);
                    if (old == w) { ++mine; break; }
                }
            }
        }
        for (int o = 32; o; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o, 64);
        if ((tid & 63) == 0) atomicAdd(&s_cnt, mine);
        __syncthreads();
        if (tid == 0) out[q] = s_fail ? 0xffffffffu : s_cnt;
        __syncthreads();
    }
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 22000, nq = argc > 2 ? atoi(argv[2]) : 10000, nd = 10000000, tbits = 15;
    const uint32_t logcap = (n + 63) / 64 * 64;
    std::vector<uint32_t> h((size_t)nq * logcap), hn(nq, n), want(nq);
    std::mt19937 rng(7);
    for (uint32_t q = 0; q < nq; ++q) {
        uint32_t *l = h.data() + (size_t)q * logcap;
        for (uint32_t i = 0; i < n; ++i) l[i] = (i > 100 && rng() % 50 == 0) ? l[rng() % i] : rng() % nd;
        if (q < 64) { std::set<uint32_t> s(l, l + n); want[q] = (uint32_t)s.size(); }
    }
    uint32_t *d_log, *d_n, *d_out;
    CK(hipMalloc(&d_log, h.size() * 4)); CK(hipMalloc(&d_n, nq * 4)); CK(hipMalloc(&d_out, nq * 4));
    CK(hipMemcpy(d_log, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_n, hn.data(), nq * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char *name, auto kern, int threads = 512) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 << tbits));
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            CK(hipMemset(d_out, 0, nq * 4));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3(256), dim3(threads), (size_t)4 << tbits, 0, d_log, logcap, d_n, nq, d_out, tbits);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
        }
        std::vector<uint32_t> o(nq); CK(hipMemcpy(o.data(), d_out, nq * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (uint32_t q = 0; q < 64 && q < nq; ++q) bad += o[q] != want[q];
        printf("{\"variant\": \"%s\", \"threads\": %d, \"n\": %u, \"nq\": %u, \"ms\": %.3f, \"wrong_of_64\": %d}\n", name, threads, n, nq, best, bad);
    };
    run("0 atomicCAS (product)", k4<0>);
    run("2 loads + clear only", k4<2>);
    run("0 atomicCAS (product)", k4<0>, 1024);
    run("4 bucket4 UN=8", k4_bucket<8>, 1024);
    run("4 bucket4 UN=4", k4_bucket<4>, 1024);
    {
        uint32_t *d_work; CK(hipMalloc(&d_work, 4));
        auto kern = k4_half;
        const size_t lds = (32768 + 4096) * 4;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            CK(hipMemset(d_out, 0, nq * 4)); CK(hipMemset(d_work, 0, 4));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, 0, d_log, logcap, d_n, nq, d_out, 24u, d_work);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
        }
        std::vector<uint32_t> o(nq); CK(hipMemcpy(o.data(), d_out, nq * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (uint32_t q = 0; q < 64 && q < nq; ++q) bad += o[q] != want[q];
        printf("{\"variant\": \"6 half-word buckets + overflow\", \"threads\": 1024, \"n\": %u, \"nq\": %u, \"ms\": %.3f, \"wrong_of_64\": %d}\n", n, nq, best, bad);
    }
    return 0;
}
