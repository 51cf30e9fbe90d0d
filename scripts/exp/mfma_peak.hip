// What does v_mfma_f32_32x32x2_f32 sustain on this chip with nothing else in the loop?  (experiment, not product)
// Register-only loop: 4 independent accumulators, operands from registers; data = random (power-realistic) or zeros.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak scripts/exp/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int WPS>
__global__ void __launch_bounds__(256, WPS) mfma_loop(const float *in, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + i + 2048) & 4095]; }
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + n) & 15], acc[n], 0, 0, 0);
    }
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum + lane;
}

int main() {
    float *d_in, *d_out;
    CK(hipMalloc(&d_in, 4096 * 4)); CK(hipMalloc(&d_out, 2048 * 256 * 4));
    float h[4096];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rnd = 0; rnd < 2; ++rnd) {
        for (int i = 0; i < 4096; ++i) h[i] = rnd ? (float)rand() / RAND_MAX - 0.5f : 0.f;
        CK(hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice));
        for (int wps = 1; wps <= 2; ++wps) {
            const int grid = 256 * wps, iters = 40000;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (wps == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, d_in, d_out, iters);
                else hipLaunchKernelGGL(mfma_loop<2>, dim3(grid), dim3(256), 0, 0, d_in, d_out, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double flop = (double)grid * 4 * iters * 64.0 * 4096.0;
                if (rep) printf("{\"data\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.2f, \"TFLOPs\": %.1f, \"frac_of_157.3\": %.3f}\n",
                                rnd ? "random" : "zeros", wps, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
            }
        }
    }
    return 0;
}
