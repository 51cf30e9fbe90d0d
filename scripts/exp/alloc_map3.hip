// alloc_map3.hip -- (same kernel as alloc_map2.hip; the tests are given on the command line)
// alloc_map2.hip -- follow-up of alloc_map.hip (round 4): device memory falls into CLASSES; K1's access mix (random rows +
// random byte tests) is 10 % slower when rows and tags sit in chunks of the same class.  Here: 2-GiB chunks over the whole
// device, each classified against a fixed set of row chunks, then compositions: rows / tags from one class, from the other,
// or striped over both.
//   hipcc --offload-arch=gfx950 -O3 -o alloc_map2 alloc_map2.hip && ./alloc_map2 [chunk GiB = 2] [max chunks = 132]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

struct Tab { uint8_t *p[48]; uint32_t n; };

// rows: row r of the logical base lives in chunk r % R.n; tags: slot s lives in chunk s % T.n
__global__ void __launch_bounds__(64) mix_kernel(Tab R, uint32_t rows_per_chunk, Tab T, size_t slot_bytes, uint32_t steps, uint32_t seed,
                                                 int do_rows, int do_tags, float *out) {
    const int lane = threadIdx.x, g = lane >> 4, a = lane & 15;
    uint8_t *my = T.p[blockIdx.x % T.n] + (size_t)(blockIdx.x / T.n) * slot_bytes;
    const uint32_t nrows = rows_per_chunk * R.n;
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
                const float *src = reinterpret_cast<const float *>(R.p[rid % R.n]) + (size_t)(rid / R.n) * 192 + a;
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
    const double ch_gib = argc > 1 ? atof(argv[1]) : 2.0;
    const int maxch = argc > 2 ? atoi(argv[2]) : 132;
    const size_t ch_bytes = (size_t)(ch_gib * (double)(1ull << 30));
    const uint32_t slots = 2048;
    const uint32_t rows_per_chunk = (uint32_t)(ch_bytes / 768);
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
    const int n = (int)ch.size();
    printf("{\"chunks\": %d, \"chunk_GiB\": %.1f}\n", n, ch_gib);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t steps = 400;
    auto run = [&](const std::vector<int> &rows, const std::vector<int> &tags, int do_rows, int do_tags) {
        Tab R{}, T{};
        R.n = (uint32_t)rows.size(); T.n = (uint32_t)tags.size();
        for (size_t i = 0; i < rows.size(); ++i) R.p[i] = ch[rows[i]];
        for (size_t i = 0; i < tags.size(); ++i) T.p[i] = ch[tags[i]];
        const uint32_t per = (slots + T.n - 1) / T.n;
        const size_t slot_bytes = ch_bytes / per / 128 * 128;
        hipLaunchKernelGGL(mix_kernel, dim3(slots), dim3(64), 0, 0, R, rows_per_chunk, T, slot_bytes, steps, 7u, do_rows, do_tags, out);
        float sum = 0;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(mix_kernel, dim3(slots), dim3(64), 0, 0, R, rows_per_chunk, T, slot_bytes, steps, 100u + r, do_rows, do_tags, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            sum += ms;
        }
        return sum / 3;
    };
    // commands: "P a,b,c"  pairwise matrix over single chunks (rows in i, tags in j);  "M a,b,c/d,e,f"  rows in a,b,c and tags in d,e,f
    auto parse = [](const char *s) { std::vector<int> v; while (*s) { v.push_back(atoi(s)); while (*s && *s != ',') ++s; if (*s == ',') ++s; } return v; };
    for (int ai = 3; ai + 1 < argc; ai += 2) {
        const char cmd = argv[ai][0];
        if (cmd == 'P') {
            const std::vector<int> v = parse(argv[ai + 1]);
            printf("{\"pairs_over\": \"%s\", \"rows_in_i_tags_in_j_ms\": [", argv[ai + 1]);
            for (size_t i = 0; i < v.size(); ++i) {
                printf("%s[", i ? ", " : "");
                for (size_t j = 0; j < v.size(); ++j) printf("%s%.2f", j ? ", " : "", run({v[i]}, {v[j]}, 1, 1));
                printf("]");
                fflush(stdout);
            }
            printf("]}\n");
        } else if (cmd == 'M') {
            std::string a = argv[ai + 1];
            const size_t sl = a.find('/');
            const std::vector<int> rows = parse(a.substr(0, sl).c_str()), tags = parse(a.substr(sl + 1).c_str());
            const float both = run(rows, tags, 1, 1), r = run(rows, tags, 1, 0), tg = run(rows, tags, 0, 1);
            printf("{\"rows_tags\": \"%s\", \"mix_ms\": %.3f, \"rows_alone_ms\": %.3f, \"tags_alone_ms\": %.3f}\n", argv[ai + 1], both, r, tg);
        }
        fflush(stdout);
    }
    return 0;
}
