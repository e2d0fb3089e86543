// What a complex FIR costs on the vector ALU of gfx950 in two forms, at 1 / 2 / 4 waves per SIMD:
//   plain: 4 v_fmac_f32 per complex multiply-add, weights as literal operands (the form lws_skew.hip uses)
//   packed: 2 v_pk_fma_f32 per complex multiply-add (op_sel swaps re / im, neg_lo negates), weights from SGPR pairs
// 33 multiply-adds with constant weights over a ring of 12 complex values per step (three sums, as a step of the skewed
// LWS kernel), 12 steps unrolled.  Prints ns per step and wave, and checks that both forms give the same numbers.
// hipcc --offload-arch=gfx950 -O3 tools/pk_fma_rate.hip -o tools/pk_fma_rate
// NB at -O3 the SLP vectoriser packs the 'plain' form as well (both kernels come out as v_pk_fma_f32); build a second
// binary with -fno-slp-vectorize to get the 4 x v_fma_f32 form in the 'plain' column (round 3: 207 vs 187 ns at 4 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr float wre(int i) { return 0.011f * (float)(i + 1) - 0.2f; }
constexpr float wim(int i) { return 0.3f - 0.007f * (float)(i + 3); }

template <bool PK>
__device__ __forceinline__ v2f cmac(v2f acc, const float wr, const float wi, v2f x) {
    if (PK) {
        v2f a = {wr, wr}, b = {-wi, wi};
        v2f xs = __builtin_shufflevector(x, x, 1, 0);
        acc = __builtin_elementwise_fma(a, x, acc);
        acc = __builtin_elementwise_fma(b, xs, acc);
        return acc;
    }
    acc.x = __builtin_fmaf(wr, x.x, acc.x);
    acc.x = __builtin_fmaf(-wi, x.y, acc.x);
    acc.y = __builtin_fmaf(wr, x.y, acc.y);
    acc.y = __builtin_fmaf(wi, x.x, acc.y);
    return acc;
}

template <bool PK, int I>
__device__ __forceinline__ void step(v2f (&R)[12], v2f& carry) {
    v2f s0 = {0.f, 0.f}, s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 11; ++p) {
        s0 = cmac<PK>(s0, wre(p), wim(p), R[(I + p) % 12]);
        s1 = cmac<PK>(s1, wre(11 + p), wim(11 + p), R[(I + p + 1) % 12]);
        s2 = cmac<PK>(s2, wre(22 + p), wim(22 + p), R[(I + p + 2) % 12]);
    }
    // the loop-carried part: newest value from the sums, renormalised
    v2f v = s0 + s1 * 0.5f + s2 * 0.25f + carry * 0.125f;
    const float n = __builtin_amdgcn_rsqf(v.x * v.x + v.y * v.y + 1e-9f);
    carry = v * n;
    R[I % 12] = carry;
}

template <bool PK>
__global__ __launch_bounds__(256) void k(float* out, int reps) {
    v2f R[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) R[i] = v2f{__builtin_cosf(0.1f * (threadIdx.x + i)), __builtin_sinf(0.1f * (threadIdx.x + i))};
    v2f carry = {1.f, 0.f};
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        step<PK, 0>(R, carry); step<PK, 1>(R, carry); step<PK, 2>(R, carry); step<PK, 3>(R, carry);
        step<PK, 4>(R, carry); step<PK, 5>(R, carry); step<PK, 6>(R, carry); step<PK, 7>(R, carry);
        step<PK, 8>(R, carry); step<PK, 9>(R, carry); step<PK, 10>(R, carry); step<PK, 11>(R, carry);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += R[i].x + 2.f * R[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int CU = 256, reps = 2000;
    float* out;
    if (hipMalloc(&out, (size_t)CU * 4 * 256 * 4 * 2) != hipSuccess) return 1;
    float* h = (float*)malloc((size_t)CU * 4 * 256 * 4 * 2);
    for (int wps : {1, 2, 4}) {          // waves per SIMD = workgroups (4 waves) per CU
        double ns[2];
        for (int pk = 0; pk < 2; ++pk) {
            float* o = out + (size_t)pk * CU * 4 * 256;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
            for (int it = 0; it < 2; ++it) {
                (void)hipEventRecord(e0, 0);
                if (pk) hipLaunchKernelGGL(k<true>, dim3(CU * wps), dim3(256), 0, 0, o, reps);
                else hipLaunchKernelGGL(k<false>, dim3(CU * wps), dim3(256), 0, 0, o, reps);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
            }
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ns[pk] = ms * 1e6 / (reps * 12.0);
        }
        if (hipMemcpy(h, out, (size_t)CU * 4 * 256 * 4 * 2, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        double worst = 0;
        for (int i = 0; i < CU * wps * 256; ++i) worst = fmax(worst, fabs((double)h[i] - (double)h[(size_t)CU * 4 * 256 + i]));
        printf("%d waves per SIMD: ns per step (all waves of a SIMD side by side): plain %.1f, packed %.1f  -> per wave-step of a SIMD "
               "%.1f / %.1f ns; max |plain - packed| %.3g\n", wps, ns[0], ns[1], ns[0] / wps, ns[1] / wps, worst);
    }
    return 0;
}
