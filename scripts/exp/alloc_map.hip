// alloc_map.hip -- WHERE device memory comes from decides how fast K1's access mix runs (round 3: two modes 10 % apart).
// A fresh process allocates chunks of CH GiB each until the device is full (allocation order = a walk over physical
// memory), then measures
//   (1) per chunk: random 768-byte row gather inside the chunk alone, and random byte tests alone;
//   (2) pairs: rows in chunk i, tags in chunk j (K1's mix), for a set of (i, j).
//   hipcc --offload-arch=gfx950 -O3 -o alloc_map alloc_map.hip && ./alloc_map [chunk GiB = 16] [max chunks = 17]
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

// do_rows / do_tags select the parts of K1's mix: 32 random 768-B rows per step (8 passes of 4 in flight), 64 random byte
// tests (+ 32 byte marks) per step in this wave's slot of the tag chunk
__global__ void __launch_bounds__(64) mix_kernel(const float *__restrict__ rows, uint32_t nrows, uint8_t *tags, size_t slot_bytes,
                                                 uint32_t steps, uint32_t seed, int do_rows, int do_tags, float *out) {
    const int lane = threadIdx.x, g = lane >> 4, a = lane & 15;
    uint8_t *my = tags + (size_t)blockIdx.x * slot_bytes;
    float acc = 0.0f;
    uint32_t s = mix(seed ^ (blockIdx.x * 0x9E3779B1u));
    for (uint32_t it = 0; it < steps; ++it) {
        s = mix(s + it);
        if (do_tags) {
            const uint32_t t = mix(s ^ (uint32_t)lane * 0x85EBCA6Bu);
            const size_t off = (size_t)(((uint64_t)t * (uint64_t)slot_bytes) >> 32);
            const uint8_t v = __hip_atomic_load(my + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane & 1) __hip_atomic_store(my + off, (uint8_t)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += (float)v;
        }
        if (do_rows) {
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
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char **argv) {
    const double ch_gib = argc > 1 ? atof(argv[1]) : 16.0;
    const int maxch = argc > 2 ? atoi(argv[2]) : 17;
    const size_t ch_bytes = (size_t)(ch_gib * (double)(1ull << 30));
    const uint32_t slots = 2048;
    const size_t slot_bytes = ch_bytes / slots / 128 * 128;
    const uint32_t nrows = (uint32_t)(ch_bytes / 768);
    CK(hipSetDevice(0));
    float *out = nullptr;
    CK(hipMalloc(&out, 64));
    std::vector<uint8_t *> ch;
    for (int i = 0; i < maxch; ++i) {
        uint8_t *p = nullptr;
        if (hipMalloc(&p, ch_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemset(p, 0, ch_bytes));
        ch.push_back(p);
    }
    CK(hipDeviceSynchronize());
    printf("{\"chunks\": %zu, \"chunk_GiB\": %.1f}\n", ch.size(), ch_gib);
    for (size_t i = 0; i < ch.size(); ++i) printf("{\"chunk\": %zu, \"ptr\": \"%p\"}\n", i, (void *)ch[i]);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t steps = 400;
    auto run = [&](int i, int j, int do_rows, int do_tags) {
        for (int w = 0; w < 2; ++w)
            hipLaunchKernelGGL(mix_kernel, dim3(slots), dim3(64), 0, 0, (const float *)ch[i], nrows, ch[j], slot_bytes, steps, 7u + w, do_rows, do_tags, out);
        float sum = 0;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(mix_kernel, dim3(slots), dim3(64), 0, 0, (const float *)ch[i], nrows, ch[j], slot_bytes, steps, 100u + r, do_rows, do_tags, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            sum += ms;
        }
        return sum / 3;
    };
    const int n = (int)ch.size();
    for (int i = 0; i < n; ++i) {
        const float a = run(i, i, 1, 0), b = run(i, i, 0, 1);
        printf("{\"alone\": %d, \"rows_ms\": %.3f, \"tags_ms\": %.3f}\n", i, a, b);
        fflush(stdout);
    }
    // pairs: rows in i, tags in j
    const int is[] = {0, 1, n / 2, n - 1};
    for (int ii = 0; ii < 4; ++ii) {
        const int i = is[ii];
        if (i < 0 || i >= n) continue;
        printf("{\"rows_in\": %d, \"mix_ms_by_tag_chunk\": [", i);
        for (int j = 0; j < n; ++j) printf("%s%.3f", j ? ", " : "", j == i ? 0.0f : run(i, j, 1, 1));
        printf("]}\n");
        fflush(stdout);
    }
    return 0;
}
