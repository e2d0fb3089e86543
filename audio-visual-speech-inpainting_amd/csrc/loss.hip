// L1 loss of the inpainter and its two diagnostics in one pass (reference models.py:144-151),
// optionally emitting d(loss)/d(prediction) = sign(p - t) / n (the tf.abs gradient) for training.
// Memory-bound: three 16-byte-per-lane streams in, one out; per-wave shuffle reduction, one
// partial per workgroup, a single-workgroup second stage that sums in double (deterministic,
// no atomics).
#include "avsi_common.h"

namespace {

constexpr int LTPB = 256;
constexpr int LMAXB = 1024;  // partials: [LMAXB][5]

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// BLEND (the speaker-embedding variants, reference models.py:1006-1011,1367-1372): `p` holds the
// sequence-masked logits on entry; prediction = row_scale * target * mask + logits * (1 - mask) is
// written back over it and the sums are taken on that prediction.
template <bool BLEND>
__global__ __launch_bounds__(LTPB) void l1_partial_kernel(const float* __restrict__ t, float* __restrict__ p,
                                                          const float* __restrict__ m, int64_t n,
                                                          float* __restrict__ dpred, float gscale,
                                                          float* __restrict__ part,
                                                          const float* __restrict__ row_scale, int row_len) {
    float s_all = 0.f, s_hole = 0.f, n_hole = 0.f, s_valid = 0.f, n_valid = 0.f;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * LTPB;
    for (int64_t i = (int64_t)blockIdx.x * LTPB + threadIdx.x; i < n4; i += stride) {
        const float4 tv = reinterpret_cast<const float4*>(t)[i];
        float4 pv = reinterpret_cast<const float4*>(p)[i];
        const float4 mv = reinterpret_cast<const float4*>(m)[i];
        if (BLEND) {
            const int64_t row = (4 * i) / row_len;
            const int rem = (int)(4 * i - row * row_len);
            float rs[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) rs[k] = row_scale ? row_scale[row + (rem + k >= row_len ? 1 : 0)] : 1.f;
            pv.x = rs[0] * tv.x * mv.x + pv.x * (1.f - mv.x);
            pv.y = rs[1] * tv.y * mv.y + pv.y * (1.f - mv.y);
            pv.z = rs[2] * tv.z * mv.z + pv.z * (1.f - mv.z);
            pv.w = rs[3] * tv.w * mv.w + pv.w * (1.f - mv.w);
            reinterpret_cast<float4*>(p)[i] = pv;
        }
        const float d[4] = {pv.x - tv.x, pv.y - tv.y, pv.z - tv.z, pv.w - tv.w};
        const float mm[4] = {mv.x, mv.y, mv.z, mv.w};
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float e = fabsf(d[k]);
            s_all += e;
            s_hole += e * (1.f - mm[k]);
            n_hole += 1.f - mm[k];
            s_valid += e * mm[k];
            n_valid += mm[k];
            g[k] = d[k] > 0.f ? gscale : (d[k] < 0.f ? -gscale : 0.f);
        }
        if (dpred) reinterpret_cast<float4*>(dpred)[i] = make_float4(g[0], g[1], g[2], g[3]);
    }
    // scalar tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        if (BLEND) p[i] = (row_scale ? row_scale[i / row_len] : 1.f) * t[i] * m[i] + p[i] * (1.f - m[i]);
        const float d = p[i] - t[i], e = fabsf(d), mk = m[i];
        s_all += e, s_hole += e * (1.f - mk), n_hole += 1.f - mk, s_valid += e * mk, n_valid += mk;
        if (dpred) dpred[i] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
    }
    __shared__ float red[LTPB / 64][5];
    float v[5] = {s_all, s_hole, n_hole, s_valid, n_valid};
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 5) {
        float s = 0.f;
        for (int w = 0; w < LTPB / 64; ++w) s += red[w][threadIdx.x];
        part[blockIdx.x * 5 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(64) void l1_final_kernel(const float* __restrict__ part, int nblocks, int64_t n,
                                                      float* __restrict__ out3, bool want_inv_hole) {
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 64)
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += (double)part[b * 5 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
    if (threadIdx.x == 0) {
        out3[0] = (float)(s[0] / (double)n);
        out3[1] = (float)(s[1] / s[2]);
        out3[2] = (float)(s[3] / s[4]);
        if (want_inv_hole) out3[3] = (float)(1.0 / s[2]);
    }
}

// d loss_hole / d logits = sign(p - t) (1 - m)^2 / sum(1 - m); the sequence mask is applied by the
// caller's relayout, like for the plain loss.  A batch without a single gap element has out4[3] = 1 / 0 = inf and
// (1 - m) = 0 everywhere: its gradient is zero, not inf * 0 (a data-parallel rank may hold such a shard while the
// global objective sum num / sum gap is well defined; the loss itself stays 0 / 0 = NaN as in the reference).
__global__ __launch_bounds__(LTPB) void hole_grad_kernel(const float* __restrict__ t, const float* __restrict__ p,
                                                         const float* __restrict__ m, int64_t n,
                                                         const float* __restrict__ out4, float* __restrict__ dlogits) {
    const float inv = isfinite(out4[3]) ? out4[3] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * LTPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * LTPB) {
        const float d = p[i] - t[i], h = 1.f - m[i];
        dlogits[i] = (d > 0.f ? inv : (d < 0.f ? -inv : 0.f)) * h * h;
    }
}

}  // namespace

// internal (frontend.hip: the loss taken inside the front-end kernel leaves one partial row per workgroup)
int avsi_l1_final_launch(const float* part, int nblocks, int64_t n, float* out3, hipStream_t st) {
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, st, part, nblocks, n, out3, false);
    return avsi_launch_status();
}

extern "C" size_t avsi_l1_loss_workspace_bytes(int64_t n) {
    (void)n;
    return (size_t)LMAXB * 5 * sizeof(float);
}

extern "C" int avsi_l1_loss_f32(const float* target, const float* pred, const float* mask, int64_t n, float* out3,
                                float* dpred, float grad_scale, void* workspace, size_t workspace_bytes, void* stream) {
    if (!target || !pred || !mask || !out3 || n <= 0) return AVSI_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < avsi_l1_loss_workspace_bytes(n)) return AVSI_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(mask) |
         reinterpret_cast<uintptr_t>(dpred)) & 15)
        return AVSI_ERR_UNSUPPORTED;
    const int64_t want = avsi_ceil_div(n >> 2, LTPB);
    const int nblocks = (int)(want < 1 ? 1 : (want > LMAXB ? LMAXB : want));
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    hipLaunchKernelGGL(l1_partial_kernel<false>, dim3(nblocks), dim3(LTPB), 0, st, target, const_cast<float*>(pred), mask,
                       n, dpred, grad_scale, (float*)workspace, (const float*)nullptr, 1);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, nblocks, n, out3, false);
    return avsi_launch_status();
}

extern "C" int avsi_l1_loss_blend_f32(const float* target, float* pred_inout, const float* mask, const float* row_scale,
                                      int row_len, int64_t n, float* out4, float* dlogits, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    if (!target || !pred_inout || !mask || !out4 || n <= 0 || row_len <= 0 || n % row_len) return AVSI_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < avsi_l1_loss_workspace_bytes(n)) return AVSI_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(pred_inout) | reinterpret_cast<uintptr_t>(mask)) & 15)
        return AVSI_ERR_UNSUPPORTED;
    const int64_t want = avsi_ceil_div(n >> 2, LTPB);
    const int nblocks = (int)(want < 1 ? 1 : (want > LMAXB ? LMAXB : want));
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    hipLaunchKernelGGL(l1_partial_kernel<true>, dim3(nblocks), dim3(LTPB), 0, st, target, pred_inout, mask, n,
                       (float*)nullptr, 0.f, (float*)workspace, row_scale, row_len);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, nblocks, n, out4, true);
    if (dlogits) {
        const int64_t gb = avsi_ceil_div(n, LTPB);
        hipLaunchKernelGGL(hole_grad_kernel, dim3((int)(gb > 4096 ? 4096 : gb)), dim3(LTPB), 0, st, target,
                           (const float*)pred_inout, mask, n, (const float*)out4, dlogits);
    }
    return avsi_launch_status();
}
