// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on gfx950 as a function of waves per SIMD
// and independent accumulators per wave.  Calibrates the roofline denominators used in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip && ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, float* d) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(d, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * (threads / 64) * iters * 8.0 * NACC * 32 * 32 * 2 * 2;
    printf("waves/SIMD %d  acc %d : %.2f ms  %.1f TFLOP/s\n", waves_per_simd, NACC, ms, flops / ms / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<1>(w, d);
        run<2>(w, d);
        run<4>(w, d);
        run<8>(w, d);
    }
    return 0;
}
