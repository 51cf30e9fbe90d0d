// alloc_place.hip -- is the speed of K1's access mix (random 768-byte rows + random byte tests / marks over ~19 GiB of tags)
// a property of HOW the tag buffer was allocated?  Round 3 found two modes 10 % apart between allocations of one buffer.
// Methods: 0 hipMalloc, 1 hipExtMallocWithFlags(hipDeviceMallocContiguous), 2 HIP VMM (one handle), 3 HIP VMM (1-GiB handles),
//          4 carved from one arena that was allocated FIRST and is never freed.
// Each method is allocated / measured / freed `trials` times, methods interleaved; optional churn (allocations of mixed
// sizes made and partly freed beforehand, the state a torch process leaves device memory in).
//   hipcc --offload-arch=gfx950 -O3 -o alloc_place alloc_place.hip && ./alloc_place [trials] [churn 0/1] [tag GiB]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// one wave per slot; per step: 64 random tag tests (one per lane), marks for half of them, then 32 random rows (4 per
// pass, lane a of a 16-lane group reads 12 dwords at a + 16 t: the compute layout of K1), eight passes in flight
__global__ void __launch_bounds__(64) mix_kernel(const float *__restrict__ rows, uint32_t nrows, uint8_t *tags, size_t slot_bytes,
                                                 uint32_t steps, uint32_t seed, float *out) {
    const int lane = threadIdx.x, g = lane >> 4, a = lane & 15;
    uint8_t *my = tags + (size_t)blockIdx.x * slot_bytes;
    float acc = 0.0f;
    uint32_t s = mix(seed ^ (blockIdx.x * 0x9E3779B1u));
    for (uint32_t it = 0; it < steps; ++it) {
        s = mix(s + it);
        const uint32_t t = mix(s ^ (uint32_t)lane * 0x85EBCA6Bu);
        const size_t off = (size_t)(((uint64_t)t * (uint64_t)slot_bytes) >> 32);
        const uint8_t v = __hip_atomic_load(my + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane & 1) __hip_atomic_store(my + off, (uint8_t)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += (float)v;
        float r[8][12];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint32_t rid = (uint32_t)(((uint64_t)mix(s ^ (uint32_t)(p * 4 + g + 1) * 0xC2B2AE35u) * nrows) >> 32);
            const float *src = rows + (size_t)rid * 192 + a;
#pragma unroll
            for (int k = 0; k < 12; ++k) r[p][k] = src[16 * k];
        }
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int k = 0; k < 12; ++k) acc += r[p][k];
    }
    if (acc == 123.456f) out[0] = acc;
}

struct Vmm { void *va = nullptr; size_t size = 0; std::vector<hipMemGenericAllocationHandle_t> h; };

static bool vmm_alloc(Vmm &v, size_t bytes, size_t chunk) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return false;
    if (chunk == 0) chunk = bytes;
    chunk = (chunk + gran - 1) / gran * gran;
    v.size = (bytes + chunk - 1) / chunk * chunk;
    if (hipMemAddressReserve(&v.va, v.size, (size_t)1 << 30, nullptr, 0) != hipSuccess) return false;
    for (size_t o = 0; o < v.size; o += chunk) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) return false;
        if (hipMemMap((char *)v.va + o, chunk, 0, h, 0) != hipSuccess) return false;
        v.h.push_back(h);
    }
    hipMemAccessDesc d = {};
    d.location = prop.location;
    d.flags = hipMemAccessFlagsProtReadWrite;
    return hipMemSetAccess(v.va, v.size, &d, 1) == hipSuccess;
}
static void vmm_free(Vmm &v) {
    if (v.va) { (void)hipMemUnmap(v.va, v.size); for (auto h : v.h) (void)hipMemRelease(h); (void)hipMemAddressFree(v.va, v.size); }
    v = Vmm();
}

int main(int argc, char **argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 5;
    const int churn = argc > 2 ? atoi(argv[2]) : 0;
    const double tag_gib = argc > 3 ? atof(argv[3]) : 19.0;
    const int arena_first = argc > 4 ? atoi(argv[4]) : 1;
    const uint32_t slots = 2048, nrows = 10000000;
    const size_t slot_bytes = (size_t)(tag_gib * (double)(1ull << 30) / slots) / 128 * 128;
    const size_t tag_bytes = slot_bytes * slots;
    CK(hipSetDevice(0));
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    printf("{\"free_GiB\": %.1f, \"total_GiB\": %.1f, \"tag_GiB\": %.2f, \"churn\": %d}\n", fr / 1073741824.0, tot / 1073741824.0, tag_bytes / 1073741824.0, churn);
    uint8_t *arena = nullptr;
    if (arena_first) CK(hipMalloc(&arena, tag_bytes));          // method 4: first thing the process allocates
    std::vector<void *> junk;
    if (churn) {     // what a torch setup phase leaves behind: blocks of mixed sizes, every other one freed
        const size_t sz[] = {(size_t)2 << 30, (size_t)512 << 20, (size_t)20 << 20, (size_t)3 << 30, (size_t)64 << 20, (size_t)1 << 30, (size_t)200 << 20};
        for (int r = 0; r < 12; ++r)
            for (size_t b : sz) { void *p = nullptr; if (hipMalloc(&p, b) == hipSuccess) junk.push_back(p); }
        for (size_t i = 0; i < junk.size(); i += 2) { (void)hipFree(junk[i]); junk[i] = nullptr; }
    }
    float *rows = nullptr, *out = nullptr;
    CK(hipMalloc(&rows, (size_t)nrows * 192 * 4));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(rows, 0, (size_t)nrows * 192 * 4));
    if (!arena_first) CK(hipMalloc(&arena, tag_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t steps = 400;
    const double bytes = (double)slots * steps * 32 * 768;
    const char *names[] = {"hipMalloc", "contiguous", "vmm_one", "vmm_1GiB", "arena_first"};
    for (int t = 0; t < trials; ++t) {
        for (int m = 0; m < 5; ++m) {
            uint8_t *tags = nullptr;
            Vmm v;
            bool ok = true;
            if (m == 0) ok = hipMalloc(&tags, tag_bytes) == hipSuccess;
            else if (m == 1) ok = hipExtMallocWithFlags((void **)&tags, tag_bytes, hipDeviceMallocContiguous) == hipSuccess;
            else if (m == 2) { ok = vmm_alloc(v, tag_bytes, 0); tags = (uint8_t *)v.va; }
            else if (m == 3) { ok = vmm_alloc(v, tag_bytes, (size_t)1 << 30); tags = (uint8_t *)v.va; }
            else tags = arena;
            if (!ok) { (void)hipGetLastError(); printf("{\"method\": \"%s\", \"trial\": %d, \"error\": \"allocation failed\"}\n", names[m], t); if (m == 2 || m == 3) vmm_free(v); continue; }
            CK(hipMemset(tags, 0, tag_bytes));
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(mix_kernel, dim3(slots), dim3(64), 0, 0, rows, nrows, tags, slot_bytes, steps, 7u + w, out);
            CK(hipDeviceSynchronize());
            float best = 1e30f, sum = 0.0f;
            for (int r = 0; r < 4; ++r) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(mix_kernel, dim3(slots), dim3(64), 0, 0, rows, nrows, tags, slot_bytes, steps, 100u + r, out);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best; sum += ms;
            }
            printf("{\"method\": \"%s\", \"trial\": %d, \"ptr\": \"%p\", \"ms_avg\": %.3f, \"ms_min\": %.3f, \"rows_TBps\": %.3f}\n", names[m], t, (void *)tags, sum / 4, best,
                   bytes / (sum / 4 * 1e-3) / 1e12);
            fflush(stdout);
            if (m == 0 || m == 1) CK(hipFree(tags));
            else if (m == 2 || m == 3) vmm_free(v);
        }
    }
    return 0;
}
