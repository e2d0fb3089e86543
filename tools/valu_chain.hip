// Issue cost of dependent vs independent VALU work for ONE wave on a SIMD (gfx950), in clock64() ticks per instruction:
//   a chain of dependent v_pk_fma_f32, four interleaved independent chains, a chain of v_fma_f32, a v_rsq_f32 + v_add_f32
//   pair -- with 64 lanes and with 1 lane active.  The tick rate is calibrated against HIP events.
// hipcc --offload-arch=gfx950 -O3 tools/valu_chain.hip -o tools/valu_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int REPS = 20000;
__global__ void k(float* out, long long* cyc, int lanes) {
    v2f a = {1.0001f, 0.9999f}, b = {1.f + threadIdx.x * 1e-7f, 1.f}, c = {1e-6f, 2e-6f};
    v2f x0 = a, x1 = b, x2 = c, x3 = a + b;
    float s = a.x, r = 2.f + threadIdx.x;
    if ((int)threadIdx.x >= lanes) return;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < REPS; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x0 = __builtin_elementwise_fma(x0, a, c);
    }
    long long t1 = clock64();
#pragma unroll 1
    for (int i = 0; i < REPS; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x0 = __builtin_elementwise_fma(x0, a, c);
            x1 = __builtin_elementwise_fma(x1, a, c);
            x2 = __builtin_elementwise_fma(x2, a, c);
            x3 = __builtin_elementwise_fma(x3, a, c);
        }
    }
    long long t2 = clock64();
#pragma unroll 1
    for (int i = 0; i < REPS; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) s = __builtin_fmaf(s, a.x, c.x);
    }
    long long t3 = clock64();
#pragma unroll 1
    for (int i = 0; i < REPS; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) r = __builtin_amdgcn_rsqf(r) + 1.5f;
    }
    long long t4 = clock64();
    out[threadIdx.x] = x0.x + x0.y + x1.x + x2.y + x3.x + s + r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0, cyc[1] = t2 - t1, cyc[2] = t3 - t2, cyc[3] = t4 - t3;
}
// many waves spinning so that the chip leaves its idle clocks before the measurement
__global__ void warm(float* out) {
    float s = threadIdx.x;
    for (int i = 0; i < 2000000; ++i) s = __builtin_fmaf(s, 1.0001f, 1e-6f);
    if (s == 12345.f) out[0] = s;
}
int main() {
    float* out;
    long long* cyc;
    if (hipMalloc(&out, 64 * 4) != hipSuccess || hipMalloc(&cyc, 64) != hipSuccess) return 1;
    hipLaunchKernelGGL(warm, dim3(1024), dim3(256), 0, 0, out);
    for (int lanes : {64, 1, 64, 1}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, lanes);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, lanes);
        (void)hipEventRecord(e1, 0);
        long long h[4];
        if (hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = 16.0 * REPS;
        printf("lanes %2d: kernel %.0f us = %lld ticks (%.0f MHz); per instruction: dependent pk_fma %.2f, 4 independent pk_fma chains %.2f, "
               "dependent fma %.2f, rsq+add pair %.2f\n", lanes, ms * 1e3, h[0] + h[1] + h[2] + h[3],
               (h[0] + h[1] + h[2] + h[3]) / (ms * 1e3), h[0] / n, h[1] / n, h[2] / n, h[3] / n);
    }
    // the same with the rest of the chip busy on another stream (does a lone wave run faster on a loaded chip?)
    hipStream_t side;
    (void)hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    for (int heaters : {64, 224}) {
        hipLaunchKernelGGL(warm, dim3(heaters * 4), dim3(256), 0, side, out + 1);      // ~4 resident workgroups per CU
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 64);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 64);
        (void)hipEventRecord(e1, 0);
        long long h[4];
        if (hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = 16.0 * REPS;
        printf("beside %d x 4 spinning workgroups: kernel %.0f us; per instruction: dependent pk_fma %.2f, 4 independent pk_fma chains %.2f, "
               "dependent fma %.2f, rsq+add pair %.2f\n", heaters, ms * 1e3, h[0] / n, h[1] / n, h[2] / n, h[3] / n);
        (void)hipStreamSynchronize(side);
    }
    return 0;
}
