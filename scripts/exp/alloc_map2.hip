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
    // classes: every chunk as the only tag chunk against rows in chunks 0..3
    const std::vector<int> ref = {0, 1, 2, 3};
    std::vector<float> t(n, 0.0f);
    printf("{\"class_ms_vs_rows_0_3\": [");
    for (int j = 4; j < n; ++j) { t[j] = run(ref, {j}, 1, 1); printf("%s%.2f", j > 4 ? ", " : "", t[j]); }
    printf("]}\n");
    fflush(stdout);
    std::vector<float> srt(t.begin() + 4, t.end());
    std::sort(srt.begin(), srt.end());
    const float lo = srt[srt.size() / 10], hi = srt[srt.size() - 1 - srt.size() / 10], mid = 0.5f * (lo + hi);
    std::vector<int> A, B;     // A: slow against the reference rows (their class), B: fast
    for (int j = 4; j < n; ++j) (t[j] > mid ? A : B).push_back(j);
    printf("{\"lo\": %.3f, \"hi\": %.3f, \"A_slow\": %zu, \"B_fast\": %zu}\n", lo, hi, A.size(), B.size());
    printf("{\"classes\": \"");
    for (int j = 4; j < n; ++j) putchar(t[j] > mid ? 'A' : 'B');
    printf("\"}\n");
    if (A.size() < 14 || B.size() < 14) { printf("{\"note\": \"classes too uneven for the composition tests\"}\n"); return 0; }
    auto take = [](const std::vector<int> &v, size_t from, size_t k) { return std::vector<int>(v.begin() + from, v.begin() + from + k); };
    auto cat = [](std::vector<int> a, const std::vector<int> &b) { a.insert(a.end(), b.begin(), b.end()); return a; };
    // spread picks: every k-th of the class
    auto spread = [](const std::vector<int> &v, size_t k) { std::vector<int> o; for (size_t i = 0; i < k; ++i) o.push_back(v[i * v.size() / k]); return o; };
    struct Case { const char *name; std::vector<int> rows, tags; };
    std::vector<Case> cases = {
        {"rows 4A, tags 10A", take(A, 0, 4), take(A, 4, 10)},
        {"rows 4A, tags 10B", take(A, 0, 4), take(B, 0, 10)},
        {"rows 4B, tags 10B", take(B, 0, 4), take(B, 4, 10)},
        {"rows 4B, tags 10A", take(B, 0, 4), take(A, 0, 10)},
        {"rows 4A, tags 5A+5B", take(A, 0, 4), cat(take(A, 4, 5), take(B, 0, 5))},
        {"rows 2A+2B, tags 5A+5B", cat(take(A, 0, 2), take(B, 0, 2)), cat(take(A, 4, 5), take(B, 4, 5))},
        {"rows 2A+2B, tags 10A", cat(take(A, 0, 2), take(B, 0, 2)), take(A, 4, 10)},
        {"rows 2A+2B, tags 10B", cat(take(A, 0, 2), take(B, 0, 2)), take(B, 4, 10)},
        {"rows 2A+2B spread, tags 5A+5B spread", cat(spread(A, 2), spread(B, 2)), cat(spread(take(A, 1, A.size() - 1), 5), spread(take(B, 1, B.size() - 1), 5))},
        {"rows 4 spread over all, tags 10 spread over all", {4, n / 4, n / 2, 3 * n / 4}, {5, 5 + n / 11, 5 + 2 * n / 11, 5 + 3 * n / 11, 5 + 4 * n / 11, 5 + 5 * n / 11, 5 + 6 * n / 11, 5 + 7 * n / 11, 5 + 8 * n / 11, 5 + 9 * n / 11}},
        {"rows 16 spread over all, tags 40 spread over all", {}, {}},
    };
    for (int i = 0; i < 16; ++i) cases.back().rows.push_back(4 + i * (n - 4) / 16);
    for (int i = 0; i < 40; ++i) cases.back().tags.push_back(5 + i * (n - 6) / 40);
    for (auto &c : cases) {
        const float both = run(c.rows, c.tags, 1, 1), r = run(c.rows, c.tags, 1, 0), tg = run(c.rows, c.tags, 0, 1);
        printf("{\"case\": \"%s\", \"mix_ms\": %.3f, \"rows_alone_ms\": %.3f, \"tags_alone_ms\": %.3f}\n", c.name, both, r, tg);
        fflush(stdout);
    }
    return 0;
}
