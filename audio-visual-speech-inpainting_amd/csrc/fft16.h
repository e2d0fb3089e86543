// In-register 16-point complex FFT building blocks shared by the forward (frontend.hip) and inverse
// (istft.hip) STFT kernels.  A 256-point complex transform is factored 16 x 16: each lane of a
// 16-lane group runs fft16() twice with one transposition through LDS in between.
#pragma once
#include <hip/hip_runtime.h>

namespace avsi_fft {

// 8-byte aligned: an LDS array of cf is then accessed with ds_read_b64 / ds_write_b64 (one access per complex
// value, 64 banks); with the natural 4-byte alignment the compiler must use ds_read2_b32 / ds_write2_b32, two
// dword accesses that each see a stride of two banks -- SQ_LDS_BANK_CONFLICT was 2/3 of all LDS cycles of the
// front-end kernel
struct alignas(8) cf {
    float r, i;
};

__device__ __forceinline__ cf cmul(cf a, cf w) { return {a.r * w.r - a.i * w.i, a.r * w.i + a.i * w.r}; }

// forward 4-point DFT, in place, natural order
__device__ __forceinline__ void fft4(cf& x0, cf& x1, cf& x2, cf& x3) {
    const cf a{x0.r + x2.r, x0.i + x2.i}, b{x0.r - x2.r, x0.i - x2.i};
    const cf c{x1.r + x3.r, x1.i + x3.i}, d{x1.r - x3.r, x1.i - x3.i};
    x0 = {a.r + c.r, a.i + c.i};
    x2 = {a.r - c.r, a.i - c.i};
    x1 = {b.r + d.i, b.i - d.r};
    x3 = {b.r - d.i, b.i + d.r};
}

// W16^m = exp(-2 pi j m / 16), m = n1*k2 for n1,k2 in 0..3
__device__ __forceinline__ cf w16(int m) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
    switch (m) {
        case 0: return {1.f, 0.f};
        case 1: return {C1, -S1};
        case 2: return {R, -R};
        case 3: return {S1, -C1};
        case 4: return {0.f, -1.f};
        case 6: return {-R, -R};
        default: return {-C1, S1};  // m == 9
    }
}

// 16-point forward DFT in registers.  In: v[n].  Out: X[k] sits at v[pos16(k)].
__device__ __forceinline__ constexpr int pos16(int k) { return 4 * (k & 3) + (k >> 2); }

__device__ __forceinline__ void fft16(cf (&v)[16]) {
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) fft4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);  // -> y[n1][k2] at v[n1+4k2]
#pragma unroll
    for (int n1 = 1; n1 < 4; ++n1)
#pragma unroll
        for (int k2 = 1; k2 < 4; ++k2) v[n1 + 4 * k2] = cmul(v[n1 + 4 * k2], w16(n1 * k2));
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) fft4(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
}


}  // namespace avsi_fft
