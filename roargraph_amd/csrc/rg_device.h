// rg_device.h -- device-side building blocks of the gfx950 search path (wave64, LDS-DMA gather).
//
// Scoring layout ("one 16-lane group per candidate row"):
//   A wave scores 4 candidate rows per sub-pass.  Lane l = 16*g + p belongs to group g (row g) and
//   owns accumulator a = p of the reference's 16-lane AVX-512 register (include/efanna2e/distance.h:
//   180-189 IP, 42-50 L2): it sees elements a, a+16, a+32, ... of the row in increasing order, one
//   fused multiply-add each, then the 16->8->4->2->1 folds (distance.h:191-222 / 52-86) are wave
//   shuffles with xor masks 8, 4, 1, 2.  IEEE addition is commutative, so the result is bit-identical
//   to the AVX-512 path.
//
// Gather: rows are fetched HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip).
//   One instruction moves 4 rows x 256 B: lane l writes 16 B at stage + 16*l.  The 16-B slot of row
//   block element group j that lane (g,p) fetches is j = (p - 4g) & 15, i.e. each row's 256-B block is
//   rotated by 64*g bytes inside its LDS region, so that the later ds_read_b32 of element 16t+a by the
//   two groups sharing a 32-lane LDS access land on disjoint banks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rg {

constexpr uint32_t kFlagBit = 0x80000000u;  // "expanded" flag of a queue entry, kept in the id's top bit
constexpr int kWave = 64;

typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;

__device__ __forceinline__ void wave_sync() {
    // single-wave workgroups: make LDS traffic of all lanes visible to all lanes, keep the compiler from reordering
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// cross-lane moves inside a 16-lane row as DPP modifiers (a VALU move, no LDS-crossbar round trip like ds_bpermute):
//   row_ror:8 == lane ^ 8, row_ror:4 == lane +- 4 (== ^4 for data of period 8), quad_perm [1,0,3,2] == ^1, [2,3,0,1] == ^2
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
// wave-wide minimum (uniform result): 4 DPP steps inside each 16-lane row, then the 4 row results through SGPRs
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_u<0xB1>(v));
    v = min(v, dpp_u<0x4E>(v));
    v = min(v, dpp_u<0x124>(v));
    v = min(v, dpp_u<0x128>(v));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

// total order of the reference's Neighbor (include/efanna2e/neighbor.h:29-31)
__device__ __forceinline__ bool nb_less(float da, uint32_t ia, float db, uint32_t ib) {
    return da < db || (da == db && ia < ib);
}

// Issue the LDS-DMA loads of one sub-pass: row `row` (this lane's group), `dim` floats, into `stage`
// (wave-uniform LDS address, ceil(dim/64) KiB).  Inactive groups issue nothing.
__device__ __forceinline__ void gather_issue(const float *__restrict__ row, uint32_t dim, bool active,
                                             float *stage, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int jsrc = (p - 4 * g) & 15;
    const uint32_t nfull = dim >> 6, rem = dim & 63u;
    const float *src = row + 4 * jsrc;
    if (active) {
        for (uint32_t b = 0; b < nfull; ++b)
            __builtin_amdgcn_global_load_lds((glb_ptr_t *)(src + 64 * b), (lds_ptr_t *)(stage + 256 * b), 16, 0, 0);
    }
    if (rem) {
        if (active && (uint32_t)(4 * jsrc) < rem)
            __builtin_amdgcn_global_load_lds((glb_ptr_t *)(src + 64 * nfull), (lds_ptr_t *)(stage + 256 * nfull), 16, 0, 0);
    }
}

// Wait until at most `n` of this wave's vector-memory loads are still outstanding (loads retire in issue order), then
// fence the compiler.  n is wave-uniform; the immediate form of s_waitcnt needs a literal, hence the switch.
__device__ __forceinline__ void gather_wait(uint32_t n) {
#define RG_W(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    switch (n) {
        RG_W(1) RG_W(2) RG_W(3) RG_W(4) RG_W(5) RG_W(6) RG_W(7) RG_W(8) RG_W(9) RG_W(10) RG_W(11) RG_W(12)
        RG_W(13) RG_W(14) RG_W(15) RG_W(16) RG_W(17) RG_W(18) RG_W(19) RG_W(20) RG_W(21) RG_W(22) RG_W(23) RG_W(24)
        RG_W(25) RG_W(26) RG_W(27) RG_W(28) RG_W(29) RG_W(30) RG_W(31) RG_W(32) RG_W(33) RG_W(34) RG_W(35) RG_W(36)
        RG_W(37) RG_W(38) RG_W(39) RG_W(40) RG_W(41) RG_W(42) RG_W(43) RG_W(44) RG_W(45) RG_W(46) RG_W(47) RG_W(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef RG_W
    __builtin_amdgcn_wave_barrier();
}
// Same wait for `mult` passes of LPP loads each when LPP is a compile-time constant (mult <= 3: ring depth <= 4): four
// immediates instead of the 48-way switch (a dozen scalar branch instructions per wait in the generic form).
template <int LPP>
__device__ __forceinline__ void gather_wait_passes(uint32_t mult) {
    static_assert(3 * LPP <= 63, "vmcnt immediate range");
    if (mult == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (mult == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPP) : "memory");
    else if (mult == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * LPP) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * LPP) : "memory");
    __builtin_amdgcn_wave_barrier();
}
// LDS-only ordering point inside a wave (does not drain the vector-memory queue, unlike wave_sync)
__device__ __forceinline__ void lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// loads one pass (4 rows) issues: used to count outstanding loads per pass
__device__ __forceinline__ uint32_t loads_per_pass(uint32_t dim) { return (dim >> 6) + ((dim & 63u) ? 1u : 0u); }

// Consume one staged sub-pass: returns the reference's compare() value of this lane's group row against qv
// (valid in every lane of the group).  dim % 8 == 0.
template <bool L2>
__device__ __forceinline__ float gather_score(const float *stage, const float *qv, uint32_t dim, int lane) {
    const int g = lane >> 4, a = lane & 15;
    const uint32_t nfull = dim >> 6, rem = dim & 63u;
    // float offset (inside a 256-float chunk) of element 16t+a of this group's row block: 64g + a + 16((t+g)&3)
    const int o0 = 64 * g + a + 16 * ((0 + g) & 3);
    const int o1 = 64 * g + a + 16 * ((1 + g) & 3);
    const int o2 = 64 * g + a + 16 * ((2 + g) & 3);
    const int o3 = 64 * g + a + 16 * ((3 + g) & 3);
    float acc = 0.0f;
#define RG_STEP(sp, off, qi)                               \
    {                                                      \
        float v = (sp)[(off)], q = qv[(qi)];               \
        if (L2) { float t = v - q; acc = __builtin_fmaf(t, t, acc); } \
        else acc = __builtin_fmaf(v, q, acc);              \
    }
    for (uint32_t b = 0; b < nfull; ++b) {
        const float *s = stage + 256 * b;
        const float *q = qv + 64 * b + a;
        { float v0 = s[o0], v1 = s[o1], v2 = s[o2], v3 = s[o3];
          float q0 = q[0], q1 = q[16], q2 = q[32], q3 = q[48];
          if (L2) {
              float t0 = v0 - q0, t1 = v1 - q1, t2 = v2 - q2, t3 = v3 - q3;
              acc = __builtin_fmaf(t0, t0, acc); acc = __builtin_fmaf(t1, t1, acc);
              acc = __builtin_fmaf(t2, t2, acc); acc = __builtin_fmaf(t3, t3, acc);
          } else {
              acc = __builtin_fmaf(v0, q0, acc); acc = __builtin_fmaf(v1, q1, acc);
              acc = __builtin_fmaf(v2, q2, acc); acc = __builtin_fmaf(v3, q3, acc);
          } }
    }
    const float *s = stage + 256 * nfull;
    const uint32_t qb = 64 * nfull;
    const uint32_t nt = rem >> 4;
    if (nt > 0) RG_STEP(s, o0, qb + a);
    if (nt > 1) RG_STEP(s, o1, qb + 16 + a);
    if (nt > 2) RG_STEP(s, o2, qb + 32 + a);
    // 16 -> 8 (distance.h:191-192): lanes a and a^8 now hold the same s8[a&7]
    acc = acc + dpp_f<0x128>(acc);
    if (rem & 8u) {  // 8-wide tail (distance.h:194-201), element 16*nt + (a&7), applied to the folded sum
        const int x = 16 * nt + (a & 7);
        const int off = 64 * g + ((x + 16 * g) & 63);
        RG_STEP(s, off, qb + x);
    }
#undef RG_STEP
    acc = acc + dpp_f<0x124>(acc);                       // 8 -> 4 (distance.h:203-204)
    acc = acc + dpp_f<0xB1>(acc);                        // first hadd  (distance.h:221)
    acc = acc + dpp_f<0x4E>(acc);                        // second hadd (distance.h:222)
    return L2 ? acc : -acc;                              // IP returns -dot (distance.h:223)
}

// Same value, same FMA order, for a compile-time dimension with the QUERY in registers: lane a of every group holds
// q[16t + a] in qr[t] (the last register holds q[16t + (a & 7)] when DIMC % 16 == 8: the 8-wide tail).  Halves the LDS
// reads of a score and frees the query's LDS (more resident queries per CU).
template <int DIMC>
__device__ __forceinline__ void load_query_regs(const float *query, float (&qr)[(DIMC + 15) / 16], int lane) {
    constexpr int QRN = (DIMC + 15) / 16;
    const int a = lane & 15;
#pragma unroll
    for (int t = 0; t < QRN; ++t) qr[t] = query[16 * t + ((16 * t + 16 <= DIMC) ? a : (a & 7))];
}

template <bool L2, int DIMC>
__device__ __forceinline__ float gather_score_q(const float *stage, const float (&qr)[(DIMC + 15) / 16], int lane) {
    static_assert(DIMC % 8 == 0 && DIMC > 0, "dimension");
    constexpr int nfull = DIMC >> 6, rem = DIMC & 63, nt = rem >> 4;
    const int g = lane >> 4, a = lane & 15;
    const int o0 = 64 * g + a + 16 * ((0 + g) & 3);
    const int o1 = 64 * g + a + 16 * ((1 + g) & 3);
    const int o2 = 64 * g + a + 16 * ((2 + g) & 3);
    const int o3 = 64 * g + a + 16 * ((3 + g) & 3);
    float acc = 0.0f;
#define RG_STEPQ(v_, q_)                                   \
    {                                                      \
        const float v = (v_), q = (q_);                    \
        if (L2) { const float t = v - q; acc = __builtin_fmaf(t, t, acc); } \
        else acc = __builtin_fmaf(v, q, acc);              \
    }
#pragma unroll
    for (int b = 0; b < nfull; ++b) {
        const float *s = stage + 256 * b;
        const float v0 = s[o0], v1 = s[o1], v2 = s[o2], v3 = s[o3];
        RG_STEPQ(v0, qr[4 * b + 0]);
        RG_STEPQ(v1, qr[4 * b + 1]);
        RG_STEPQ(v2, qr[4 * b + 2]);
        RG_STEPQ(v3, qr[4 * b + 3]);
    }
    const float *s = stage + 256 * nfull;
    if constexpr (nt > 0) RG_STEPQ(s[o0], qr[4 * nfull + 0]);
    if constexpr (nt > 1) RG_STEPQ(s[o1], qr[4 * nfull + 1]);
    if constexpr (nt > 2) RG_STEPQ(s[o2], qr[4 * nfull + 2]);
    acc = acc + dpp_f<0x128>(acc);                       // 16 -> 8
    if constexpr ((rem & 8) != 0) {                      // 8-wide tail on the folded sum
        const int x = 16 * nt + (a & 7);
        const int off = 64 * g + ((x + 16 * g) & 63);
        RG_STEPQ(s[off], qr[(DIMC + 15) / 16 - 1]);
    }
#undef RG_STEPQ
    acc = acc + dpp_f<0x124>(acc);
    acc = acc + dpp_f<0xB1>(acc);
    acc = acc + dpp_f<0x4E>(acc);
    return L2 ? acc : -acc;
}

// The same value again for a pass that sits in REGISTERS (rg_search_kernel's register-staged gather): rv[b] is the 16 bytes
// this lane fetched of 64-element block b of its group's row -- elements 64b + 4*((p - 4g) & 15) .. +3, the slot LDS-DMA
// would have filled.  Block by block the wave drops its 1 KiB into ONE 1-KiB LDS buffer and reads it back transposed
// (lane a gets elements a, a+16, a+32, a+48 of its row's block).  The LDS unit executes a wave's DS instructions in issue
// order, so block b+1 may be written behind the reads of block b without waiting for their data; only the FMAs wait.
// One KiB per in-flight query instead of ceil(dim/64): at wide beams LDS is what limits resident queries.
typedef float v4f_t __attribute__((ext_vector_type(4)));
// The 8-element remainder of a d = 200 row needs no bounce: after the 16 -> 8 fold lane a applies element 192 + (a & 7)
// (distance.h:194-201), so every lane fetches that one float itself (`t8`; from behind the row or, with split rows, from the
// per-edge tail array).
template <bool L2, int DIMC>
__device__ __forceinline__ float bounce_score_q(float *stage1k, const v4f_t (&rv)[DIMC / 64], float t8, const float (&qr)[(DIMC + 15) / 16],
                                                int lane) {
    static_assert(DIMC % 8 == 0 && DIMC > 0, "dimension");
    constexpr int nfull = DIMC >> 6, rem = DIMC & 63;
    static_assert(rem == 0 || rem == 8, "register-staged gather: whole 64-element blocks plus at most the 8-wide tail");
    const int g = lane >> 4, a = lane & 15;
    const int o0 = 64 * g + a + 16 * ((0 + g) & 3);
    const int o1 = 64 * g + a + 16 * ((1 + g) & 3);
    const int o2 = 64 * g + a + 16 * ((2 + g) & 3);
    const int o3 = 64 * g + a + 16 * ((3 + g) & 3);
    float acc = 0.0f;
#define RG_STEPB(v_, q_)                                   \
    {                                                      \
        const float v = (v_), q = (q_);                    \
        if (L2) { const float t = v - q; acc = __builtin_fmaf(t, t, acc); } \
        else acc = __builtin_fmaf(v, q, acc);              \
    }
#pragma unroll
    for (int b = 0; b < nfull; ++b) {
        *reinterpret_cast<v4f_t *>(stage1k + 4 * lane) = rv[b];
        asm volatile("" ::: "memory");
        const float v0 = stage1k[o0], v1 = stage1k[o1], v2 = stage1k[o2], v3 = stage1k[o3];
        asm volatile("" ::: "memory");
        RG_STEPB(v0, qr[4 * b + 0]);
        RG_STEPB(v1, qr[4 * b + 1]);
        RG_STEPB(v2, qr[4 * b + 2]);
        RG_STEPB(v3, qr[4 * b + 3]);
    }
    acc = acc + dpp_f<0x128>(acc);                       // 16 -> 8
    if constexpr (rem == 8) RG_STEPB(t8, qr[(DIMC + 15) / 16 - 1]);   // 8-wide tail on the folded sum
#undef RG_STEPB
    acc = acc + dpp_f<0x124>(acc);
    acc = acc + dpp_f<0xB1>(acc);
    acc = acc + dpp_f<0x4E>(acc);
    return L2 ? acc : -acc;
}

// The same value once more for a row fetched in the COMPUTE layout (round 3, gather form 1): lane a of the row's group
// loaded elements a, a + 16, a + 32, ... itself (one dword each: every load instruction of the wave reads four 64-byte
// segments, two consecutive instructions cover a row's 128-byte line), so rv[t] IS the operand of step t and the score
// is twelve (d = 200) / thirty-two (d = 512) FMAs on registers -- no LDS bounce, no transposition.  Same FMA order.
template <bool L2, int DIMC>
__device__ __forceinline__ float regs_score_q(const float (&rv)[DIMC / 16], float t8, const float (&qr)[(DIMC + 15) / 16]) {
    static_assert(DIMC % 8 == 0 && DIMC > 0, "dimension");
    constexpr int NT = DIMC / 16, rem = DIMC & 15;
    float acc = 0.0f;
#define RG_STEPR(v_, q_)                                   \
    {                                                      \
        const float v = (v_), q = (q_);                    \
        if (L2) { const float t = v - q; acc = __builtin_fmaf(t, t, acc); } \
        else acc = __builtin_fmaf(v, q, acc);              \
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) RG_STEPR(rv[t], qr[t]);
    acc = acc + dpp_f<0x128>(acc);                       // 16 -> 8
    if constexpr (rem == 8) RG_STEPR(t8, qr[(DIMC + 15) / 16 - 1]);   // 8-wide tail on the folded sum
#undef RG_STEPR
    acc = acc + dpp_f<0x124>(acc);
    acc = acc + dpp_f<0xB1>(acc);
    acc = acc + dpp_f<0x4E>(acc);
    return L2 ? acc : -acc;
}

// ---- opt-in fast mode (NOT parity): traversal over a bf16 copy of the base ---------------------------
// Rows of the copy are padded with zeros to a multiple of 128 elements (256 B: whole LDS-DMA instructions, whole
// 128-B lines: 512 B per d = 200 row instead of the 7 lines = 896 B of the fp32 row).  One 16-lane group per row as in
// the exact path; lane p owns the 8 elements 128b + 8p .. +7 of every 128-element block b and reads back exactly the
// 16 bytes it fetched.
template <int NB>
__device__ __forceinline__ void gather_issue_bf(const uint16_t *__restrict__ row, bool active, uint32_t *stage, int lane) {
    const int p = lane & 15;
    if (active) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
            __builtin_amdgcn_global_load_lds((glb_ptr_t *)(row + 128 * b + 8 * p), (lds_ptr_t *)(stage + 256 * b), 16, 0, 0);
    }
}

// query fragment of the fast mode: qb[8b + i] = q[128b + 8p + i] (0 beyond the dimension)
template <int DIMC>
__device__ __forceinline__ void load_query_regs_bf(const float *query, float (&qb)[8 * ((DIMC + 127) / 128)], int lane) {
    constexpr int NB = (DIMC + 127) / 128;
    const int p = lane & 15;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = 128 * b + 8 * p + i;
            qb[8 * b + i] = e < DIMC ? query[e] : 0.0f;
        }
}

// approximate compare() of this lane's group row (bf16 elements, fp32 query, fp32 accumulation); valid in every lane
template <bool L2, int NB>
__device__ __forceinline__ float score_bf(const uint32_t *stage, const float (&qb)[8 * NB], int lane) {
    float acc = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint4 w = *reinterpret_cast<const uint4 *>(stage + 256 * b + 4 * lane);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float lo = __uint_as_float(ww[i] << 16), hi = __uint_as_float(ww[i] & 0xffff0000u);
            const float q0 = qb[8 * b + 2 * i], q1 = qb[8 * b + 2 * i + 1];
            if (L2) {
                const float t0 = lo - q0, t1 = hi - q1;
                acc = __builtin_fmaf(t0, t0, acc);
                acc = __builtin_fmaf(t1, t1, acc);
            } else {
                acc = __builtin_fmaf(lo, q0, acc);
                acc = __builtin_fmaf(hi, q1, acc);
            }
        }
    }
    acc = acc + dpp_f<0x128>(acc);
    acc = acc + dpp_f<0x124>(acc);
    acc = acc + dpp_f<0xB1>(acc);
    acc = acc + dpp_f<0x4E>(acc);
    return L2 ? acc : -acc;
}

}  // namespace rg
