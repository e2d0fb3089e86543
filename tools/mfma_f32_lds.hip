// Microbenchmark: v_mfma_f32_32x32x2_f32 fed from LDS the way gemm.hip's main loop does it
// (per 16 MFMAs: 2 x ds_read_b128 + 8 x ds_read_b32), no barriers, no global traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: registers only, 1: LDS reads one group ahead, 2: LDS reads right before use,
                      // 3: mode 1 + barrier every 32 MFMAs, 4: mode 3 + 4 ds_write_b128 per barrier, 5: mode 4 + 4 global float4 loads
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, const float4* __restrict__ gsrc) {
    __shared__ __attribute__((aligned(16))) float sa[128 * 36], sb[32 * 132], sdma[16 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
    for (int i = tid; i < 128 * 36; i += 256) sa[i] = 1e-3f * (i % 97);
    for (int i = tid; i < 32 * 132; i += 256) sb[i] = 1e-3f * (i % 89);
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float af[2][2][4], bf[2][2][4];
    auto rd = [&](float (&a)[2][4], float (&b)[2][4], int q) {
        for (int i = 0; i < 2; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(sa + (wm * 64 + i * 32 + li) * 36 + 8 * q + 4 * hi);
            a[i][0] = v.x, a[i][1] = v.y, a[i][2] = v.z, a[i][3] = v.w;
        }
        for (int j = 0; j < 2; ++j)
            for (int s = 0; s < 4; ++s) b[j][s] = sb[(8 * q + 4 * hi + s) * 132 + wn * 64 + j * 32 + li];
    };
    rd(af[0], bf[0], 0);
    float4 stage[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cq = q & 1;
            if (MODE >= 3 && (q & 1) == 0) {
                if (MODE == 4 || MODE == 5) {
                    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(sa + ((p * 64 + (tid >> 2)) % 128) * 36 + 4 * (tid & 3) + 16) = stage[p];
                }
                if (MODE == 8) {   // mode 7 with the GEMM's row-tile source pattern: 16 rows x 64 B per instruction, 16-KiB row pitch
                    for (int p = 0; p < 4; ++p)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(gsrc + ((((size_t)(blockIdx.x * 8 + wave * 2 + (p & 1)) * 16 + (lane >> 2)) * 1024 + (size_t)(it * 4 + q + (p >> 1)) * 4 + (lane & 3)) & ((1u << 24) - 1))),
                            (__attribute__((address_space(3))) void*)(sdma + (wave * 4 + p) * 256), 16, 0, 0);
                }
                if (MODE == 9) {   // mode 7 with the DMA landing in the arrays the fragment reads use
                    for (int p = 0; p < 4; ++p)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(gsrc + (((size_t)(blockIdx.x * 997 + it * 8 + q * 2 + p) * 256 + tid) & ((1u << 24) - 1))),
                            (__attribute__((address_space(3))) void*)((p < 2 ? sa : sb) + (wave * 2 + (p & 1)) * 256), 16, 0, 0);
                }
                if (MODE == 6 || MODE == 7) {   // LDS-DMA: global -> LDS without VGPR staging / ds_write (4 x 1 KiB per wave)
                    for (int p = 0; p < 4; ++p)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(gsrc + (((size_t)(blockIdx.x * 997 + it * 8 + q * 2 + p) * 256 + tid) & ((1u << 24) - 1))),
                            (__attribute__((address_space(3))) void*)(sdma + (wave * 4 + p) * 256), 16, 0, 0);
                }
                if (MODE == 5) {
                    for (int p = 0; p < 4; ++p) stage[p] = gsrc[((size_t)(blockIdx.x * 997 + it * 8 + q * 2 + p) * 256 + tid) & ((1u << 24) - 1)];
                }
            }
            if (MODE == 1 || MODE >= 3) { rd(af[cq ^ 1], bf[cq ^ 1], (q + 1) & 3); __builtin_amdgcn_sched_barrier(0); }
            if (MODE == 2) { rd(af[cq], bf[cq], q); }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[MODE == 0 ? 0 : cq][i][s], bf[MODE == 0 ? 0 : cq][j][s], acc[i][j], 0, 0, 0);
            if (MODE >= 3 && MODE < 7 && (q & 1) == 1) __syncthreads();
            if (MODE >= 7 && (q & 1) == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int MODE> void run(float* d, int blocks, const float4* g) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10, g);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, g);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64.0 * 32 * 32 * 2 * 2;
    printf("mode %d blocks %d: %.2f ms  %.1f TFLOP/s\n", MODE, blocks, ms, flops / ms / 1e9);
}
int main() {
    float* d; (void)hipMalloc(&d, 2048 * 256 * 4);
    float4* g; (void)hipMalloc(&g, (size_t)(1u << 24) * 16); (void)hipMemset(g, 0, (size_t)(1u << 24) * 16);
    for (int blocks : {256, 512, 1024}) { run<0>(d, blocks, g); run<1>(d, blocks, g); run<3>(d, blocks, g); run<4>(d, blocks, g); run<5>(d, blocks, g); run<6>(d, blocks, g); run<7>(d, blocks, g); run<8>(d, blocks, g); run<9>(d, blocks, g); }
    return 0;
}
