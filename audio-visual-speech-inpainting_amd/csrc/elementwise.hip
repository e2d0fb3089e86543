// Memory-bound helpers of the training / layout plumbing on gfx950:
//   - avsi_relayout_rows_f32 : [B][T][C] <-> [T][Bp][Cp] row re-layout with optional per-row scale
//                              and zero fill (video / fed features -> padded time-major input;
//                              d(prediction) -> time-major gradient with the sequence mask folded in)
//   - avsi_colsum_f32        : column sums of a [M][ld] matrix (bias gradients), two-stage, deterministic
//   - avsi_sum_slabs_f32     : sum of split-K partial slabs (gemm.hip), deterministic
//   - avsi_adam_tf_f32       : tf.train.AdamOptimizer update on flat buffers (models.py:168; App. A.7)
//   - avsi_adam_tf_guarded_f32 / avsi_step_guard_f32 : the same update behind a device-side step guard (a non-finite loss or
//                              a cooperative-kernel timeout on any rank voids the step before a variable is touched)
// All are grid-stride kernels with 16-byte accesses where alignment allows.
#include "avsi_common.h"

namespace {

constexpr int TPB = 256;

__global__ __launch_bounds__(TPB) void relayout_rows_kernel(const float* __restrict__ src, int64_t s_sb, int64_t s_st,
                                                            float* __restrict__ dst, int64_t d_sb, int64_t d_st, int B,
                                                            int T, int C, int dst_cols, const float* __restrict__ scale,
                                                            int64_t sc_sb, int64_t sc_st) {
    // one wave per (b, t) row; lanes stride the channels
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    const int64_t rows = (int64_t)B * T;
    for (int64_t row = wave; row < rows; row += nwaves) {
        const int b = (int)(row / T), t = (int)(row - (int64_t)b * T);
        const float sc = scale ? scale[b * sc_sb + t * sc_st] : 1.f;
        const float* s = src + b * s_sb + t * s_st;
        float* d = dst + b * d_sb + t * d_st;
        for (int c = lane; c < dst_cols; c += 64) d[c] = c < C ? s[c] * sc : 0.f;
    }
}

constexpr int CS_ROWS = 256;  // rows per partial block
__global__ __launch_bounds__(TPB) void colsum_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t M, int N,
                                                             float* __restrict__ part) {
    // block (bx, by): columns [bx*256, +256), rows [by*chunk, ...)
    const int col = blockIdx.x * TPB + threadIdx.x;
    const int64_t chunk = (M + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = min(M, r0 + chunk);
    if (col >= N) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += x[r * ld + col];
        s1 += x[(r + 1) * ld + col];
        s2 += x[(r + 2) * ld + col];
        s3 += x[(r + 3) * ld + col];
    }
    for (; r < r1; ++r) s0 += x[r * ld + col];
    part[(int64_t)blockIdx.y * N + col] = (s0 + s1) + (s2 + s3);
}

// Narrow matrices (ld <= 256: bias gradients of the U-Net layers, 16 .. 128 channels): with one
// thread per column only ld threads of a block work.  Here rows are read whole with 16-byte loads
// (thread t: float4 column group t % (ld / 4), rows t / (ld / 4) + i * TPB / (ld / 4)) and the
// row-threads of a column are summed through LDS in a fixed order.
__global__ __launch_bounds__(TPB) void colsum_narrow_kernel(const float* __restrict__ x, int ld, int64_t M, int N,
                                                            float* __restrict__ part) {
    __shared__ float red[TPB * 4];
    const int q4 = ld >> 2, rp = TPB / q4;
    const int cq = threadIdx.x % q4, rl = threadIdx.x / q4;
    const int64_t chunk = (M + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = min(M, r0 + chunk);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < rp)
        for (int64_t r = r0 + rl; r < r1; r += rp) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * ld + 4 * cq);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
    *reinterpret_cast<float4*>(red + threadIdx.x * 4) = s;
    __syncthreads();
    const int c = threadIdx.x;
    if (c < N) {
        float t = 0.f;
        for (int j = 0; j < rp; ++j) t += red[(j * q4 + (c >> 2)) * 4 + (c & 3)];
        part[(int64_t)blockIdx.y * N + c] = t;
    }
}

__global__ __launch_bounds__(TPB) void sum_slabs_kernel(const float* __restrict__ slabs, int64_t n, int count,
                                                        int64_t stride, float* __restrict__ out, float alpha) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        double s = 0.0;
        for (int k = 0; k < count; ++k) s += (double)slabs[k * stride + i];
        out[i] = (float)(alpha * s);
    }
}

// The same sum with the slabs of one output spread over PT = TPB / LN threads (partition p takes slabs p, p + PT, ..., four
// loads in flight; the partitions are added in order through LDS: a fixed order, so still deterministic) and 16-byte
// accesses.  The one-thread-per-output form above walks `count` slabs serially: with few outputs (a thin layer's filter
// gradient: 6912 floats in 256 slabs) it was 27 workgroups of serial loads -- 56 us per call, 29 calls per U-Net training step.
// `out` may alias slab 0 (every read of a workgroup's columns is behind the barrier before their store).
template <int LN>
__global__ __launch_bounds__(TPB) void sum_slabs_v4_kernel(const float* slabs, int64_t n4, int count, int64_t stride4,
                                                           float* out, float alpha) {
    constexpr int PT = TPB / LN;
    __shared__ double red[PT][LN][4];
    const int l = threadIdx.x % LN, p = threadIdx.x / LN;
    for (int64_t base = (int64_t)blockIdx.x * LN; base < n4; base += (int64_t)gridDim.x * LN) {
        const int64_t i = base + l;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (i < n4) {
            const float4* src = reinterpret_cast<const float4*>(slabs) + i;
            int k = p;
            for (; k + 3 * PT < count; k += 4 * PT) {
                const float4 a = src[(int64_t)k * stride4], b = src[(int64_t)(k + PT) * stride4];
                const float4 c = src[(int64_t)(k + 2 * PT) * stride4], d = src[(int64_t)(k + 3 * PT) * stride4];
                s0 += (double)a.x, s1 += (double)a.y, s2 += (double)a.z, s3 += (double)a.w;
                s0 += (double)b.x, s1 += (double)b.y, s2 += (double)b.z, s3 += (double)b.w;
                s0 += (double)c.x, s1 += (double)c.y, s2 += (double)c.z, s3 += (double)c.w;
                s0 += (double)d.x, s1 += (double)d.y, s2 += (double)d.z, s3 += (double)d.w;
            }
            for (; k < count; k += PT) {
                const float4 a = src[(int64_t)k * stride4];
                s0 += (double)a.x, s1 += (double)a.y, s2 += (double)a.z, s3 += (double)a.w;
            }
        }
        red[p][l][0] = s0, red[p][l][1] = s1, red[p][l][2] = s2, red[p][l][3] = s3;
        __syncthreads();
        if (p == 0 && i < n4) {
            for (int q = 1; q < PT; ++q) s0 += red[q][l][0], s1 += red[q][l][1], s2 += red[q][l][2], s3 += red[q][l][3];
            reinterpret_cast<float4*>(out)[i] = make_float4((float)(alpha * s0), (float)(alpha * s1), (float)(alpha * s2),
                                                            (float)(alpha * s3));
        }
        __syncthreads();
    }
}

// Few outputs, many slabs, no 16-byte alignment (the 153 and 1 filter weights of the one-channel layers, 512 slabs): one
// workgroup per output, a slab subset per thread, a fixed shuffle / LDS tree (deterministic).  The serial form took 192 us.
__global__ __launch_bounds__(TPB) void sum_slabs_small_kernel(const float* slabs, int count, int64_t stride, float* out,
                                                              float alpha) {
    __shared__ double red[TPB / 64];
    const int64_t i = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < count; k += TPB) s += (double)slabs[k * stride + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[i] = (float)(alpha * ((red[0] + red[1]) + (red[2] + red[3])));
}

__global__ __launch_bounds__(TPB) void adam_tf_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                      float lr_t, float b1, float b2, float eps, float gscale,
                                                      float l2, const float* __restrict__ skip, int n_skip) {
    // step guard: any word that is not exactly zero (NaN included) voids the step -- nothing is read or written
    for (int k = 0; k < n_skip; ++k)
        if (!(skip[k] == 0.f)) return;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x;
        const float* gg = &gv.x;
        float* mm = &mv.x;
        float* vw = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * gscale + l2 * pp[k];
            mm[k] = b1 * mm[k] + (1.f - b1) * gk;
            vw[k] = b2 * vw[k] + (1.f - b2) * gk * gk;
            pp[k] -= lr_t * mm[k] / (__builtin_amdgcn_sqrtf(vw[k]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float gk = g[i] * gscale + l2 * p[i];
        m[i] = b1 * m[i] + (1.f - b1) * gk;
        v[i] = b2 * v[i] + (1.f - b2) * gk * gk;
        p[i] -= lr_t * m[i] / (__builtin_amdgcn_sqrtf(v[i]) + eps);
    }
}

// tf.train.GradientDescentOptimizer / tf.train.MomentumOptimizer(momentum) behind the same step guard (models.py:170-176):
// g' = g * gscale + l2 * p;  plain: p -= lr * g';  momentum: accum = momentum * accum + g', p -= lr * accum (TF's form: the
// learning rate multiplies the accumulator at apply time, it is not folded into it; no Nesterov term -- the reference passes none).
template <bool MOMENTUM>
__global__ __launch_bounds__(TPB) void sgd_tf_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ accum, int64_t n, float lr, float momentum,
                                                     float gscale, float l2, const float* __restrict__ skip, int n_skip) {
    for (int k = 0; k < n_skip; ++k)
        if (!(skip[k] == 0.f)) return;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 av = MOMENTUM ? reinterpret_cast<float4*>(accum)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float* pp = &pv.x;
        const float* gg = &gv.x;
        float* aa = &av.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * gscale + l2 * pp[k];
            if (MOMENTUM) {
                aa[k] = momentum * aa[k] + gk;
                pp[k] -= lr * aa[k];
            } else {
                pp[k] -= lr * gk;
            }
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        if (MOMENTUM) reinterpret_cast<float4*>(accum)[i] = av;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float gk = g[i] * gscale + l2 * p[i];
        if (MOMENTUM) {
            accum[i] = momentum * accum[i] + gk;
            p[i] -= lr * accum[i];
        } else {
            p[i] -= lr * gk;
        }
    }
}

// out[0] = NaN unless *loss is finite, out[1] = 1 if a cooperative status word is set (else 0): the two words of the
// step guard, summed over the data-parallel ranks inside the last gradient bucket
__global__ void step_guard_kernel(const float* loss, const int* status_a, const int* status_b, float* out) {
    if (threadIdx.x == 0) {
        const bool finite = !loss || (*loss - *loss == 0.f);
        out[0] = finite ? 0.f : __builtin_nanf("");
        out[1] = ((status_a && *status_a != 0) || (status_b && *status_b != 0)) ? 1.f : 0.f;
    }
}

inline int grid_for(int64_t items, int per_block) {
    int64_t g = avsi_ceil_div(items, per_block);
    const int64_t cap = (int64_t)AVSI_NUM_CU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int avsi_relayout_rows_f32(const float* src, int64_t src_stride_b, int64_t src_stride_t, float* dst,
                                      int64_t dst_stride_b, int64_t dst_stride_t, int B, int T, int C, int dst_cols,
                                      const float* row_scale, int64_t scale_stride_b, int64_t scale_stride_t,
                                      void* stream) {
    if (!src || !dst || B <= 0 || T <= 0 || C <= 0 || dst_cols < C) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(relayout_rows_kernel, dim3(grid_for((int64_t)B * T, TPB / 64)), dim3(TPB), 0,
                       (hipStream_t)stream, src, src_stride_b, src_stride_t, dst, dst_stride_b, dst_stride_t, B, T, C,
                       dst_cols, row_scale, scale_stride_b, scale_stride_t);
    return avsi_launch_status();
}

extern "C" size_t avsi_colsum_workspace_bytes(int64_t M, int N) {
    const int64_t parts = avsi_ceil_div(M, CS_ROWS) > 512 ? 512 : avsi_ceil_div(M, CS_ROWS);
    return (size_t)(parts < 1 ? 1 : parts) * N * sizeof(float);
}

extern "C" int avsi_colsum_f32(const float* x, int64_t ld, int64_t M, int N, float* out, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!x || !out || M <= 0 || N <= 0 || ld < N) return AVSI_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < avsi_colsum_workspace_bytes(M, N)) return AVSI_ERR_WORKSPACE;
    int64_t parts = avsi_ceil_div(M, CS_ROWS);
    if (parts > 512) parts = 512;
    if (parts < 1) parts = 1;
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    if (ld <= 256 && !(ld & 3) && !(reinterpret_cast<uintptr_t>(x) & 15))
        hipLaunchKernelGGL(colsum_narrow_kernel, dim3(1, (int)parts), dim3(TPB), 0, st, x, (int)ld, M, N, (float*)workspace);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((int)avsi_ceil_div(N, TPB), (int)parts), dim3(TPB), 0, st, x, ld, M,
                           N, (float*)workspace);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(grid_for(N, TPB)), dim3(TPB), 0, st, (const float*)workspace, (int64_t)N,
                       (int)parts, (int64_t)N, out, 1.f);
    return avsi_launch_status();
}

// tf.nn.dropout(x, rate) (reference models.py:117, on the last BLSTM layer's output in front of the projection):
// y = x * scale, scale = 0 with probability rate, 1 / (1 - rate) otherwise.  The draw is a counter-based generator
// (splitmix64 of seed + element index): reproducible for a given (seed, shape), independent of the launch geometry.
// TensorFlow's own random stream cannot be reproduced, so `scale` is written out: the backward pass multiplies by it
// and tests feed the same factors to the oracle.  Rows are [rows][ld] with `cols` live columns.
namespace {
__global__ __launch_bounds__(TPB) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      float* __restrict__ scale, int64_t rows, int cols, int ld, float rate,
                                                      unsigned long long seed) {
    const int64_t n = rows * cols;
    const float keep_scale = 1.f / (1.f - rate);
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int64_t r = e / cols;
        const int c = (int)(e - r * cols);
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(e + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const float u = (float)(z >> 40) * (1.f / 16777216.f);          // 24 uniform bits in [0, 1)
        const float sc = u >= rate ? keep_scale : 0.f;
        const int64_t o = r * ld + c;
        scale[o] = sc;
        y[o] = x[o] * sc;
    }
}
__global__ __launch_bounds__(TPB) void scale_rows_kernel(float* __restrict__ x, const float* __restrict__ scale, int64_t rows,
                                                         int cols, int ld) {
    const int64_t n = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int64_t r = e / cols;
        const int64_t o = r * ld + (e - r * cols);
        x[o] *= scale[o];
    }
}
}  // namespace

extern "C" int avsi_dropout_f32(const float* x, float* y, float* scale, int64_t rows, int cols, int ld, float rate,
                                unsigned long long seed, void* stream) {
    if (!x || !y || !scale || rows <= 0 || cols <= 0 || ld < cols || !(rate >= 0.f) || !(rate < 1.f)) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(rows * cols, TPB)), dim3(TPB), 0, (hipStream_t)stream, x, y, scale, rows, cols,
                       ld, rate, seed);
    return avsi_launch_status();
}

// x[r][c] *= scale[r][c] on the live columns (the gradient of the dropout above)
extern "C" int avsi_scale_elements_f32(float* x, const float* scale, int64_t rows, int cols, int ld, void* stream) {
    if (!x || !scale || rows <= 0 || cols <= 0 || ld < cols) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for(rows * cols, TPB)), dim3(TPB), 0, (hipStream_t)stream, x, scale, rows, cols, ld);
    return avsi_launch_status();
}

// internal (gemm.hip): out[i] = alpha * sum_k slabs[k * stride + i]
int avsi_sum_slabs_launch(const float* slabs, int64_t n, int count, int64_t stride, float* out, float alpha,
                          hipStream_t st) {
    if (count >= 2 && !(n & 3) && !(stride & 3) &&
        !((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out)) & 15)) {
        const int64_t n4 = n >> 2;
        if (n4 >= 32768)
            hipLaunchKernelGGL(sum_slabs_v4_kernel<64>, dim3((unsigned)avsi_ceil_div(n4, (int64_t)64)), dim3(TPB), 0, st, slabs, n4,
                               count, stride >> 2, out, alpha);
        else
            hipLaunchKernelGGL(sum_slabs_v4_kernel<16>, dim3((unsigned)avsi_ceil_div(n4, (int64_t)16)), dim3(TPB), 0, st, slabs, n4,
                               count, stride >> 2, out, alpha);
        return avsi_launch_status();
    }
    if (n <= 2048 && count >= 64) {      // (out may be slab 0: a workgroup reads its column of every slab before it writes it)
        hipLaunchKernelGGL(sum_slabs_small_kernel, dim3((unsigned)n), dim3(TPB), 0, st, slabs, count, stride, out, alpha);
        return avsi_launch_status();
    }
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(grid_for(n, TPB)), dim3(TPB), 0, st, slabs, n, count, stride, out, alpha);
    return avsi_launch_status();
}

extern "C" int avsi_step_guard_f32(const float* loss, const int* status_a, const int* status_b, float* out2, void* stream) {
    if (!out2) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(step_guard_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss, status_a, status_b, out2);
    return avsi_launch_status();
}

extern "C" int avsi_adam_tf_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                                float beta2, float eps, int64_t step, float grad_scale, float l2, void* stream) {
    return avsi_adam_tf_guarded_f32(param, grad, m, v, n, lr, beta1, beta2, eps, step, grad_scale, l2, nullptr, 0, stream);
}

extern "C" int avsi_adam_tf_guarded_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                                        float beta2, float eps, int64_t step, float grad_scale, float l2, const float* skip,
                                        int n_skip, void* stream) {
    if (!param || !grad || !m || !v || n <= 0 || step < 1) return AVSI_ERR_INVALID_ARG;
    if (n_skip < 0 || n_skip > 8 || (n_skip > 0 && !skip)) return AVSI_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return AVSI_ERR_UNSUPPORTED;
    // lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), computed in double on the host
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    avsi_clear_error();
    hipLaunchKernelGGL(adam_tf_kernel, dim3(grid_for(n >> 2, TPB)), dim3(TPB), 0, (hipStream_t)stream, param, grad, m, v,
                       n, (float)lr_t, beta1, beta2, eps, grad_scale, l2, skip, n_skip);
    return avsi_launch_status();
}

extern "C" int avsi_sgd_momentum_f32(float* param, const float* grad, float* accum, int64_t n, float lr, float momentum,
                                     float grad_scale, float l2, const float* skip, int n_skip, void* stream) {
    if (!param || !grad || n <= 0 || !(momentum >= 0.f)) return AVSI_ERR_INVALID_ARG;
    if (n_skip < 0 || n_skip > 8 || (n_skip > 0 && !skip)) return AVSI_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(accum)) & 15)
        return AVSI_ERR_UNSUPPORTED;
    avsi_clear_error();
    if (accum)
        hipLaunchKernelGGL(sgd_tf_kernel<true>, dim3(grid_for(n >> 2, TPB)), dim3(TPB), 0, (hipStream_t)stream, param, grad,
                           accum, n, lr, momentum, grad_scale, l2, skip, n_skip);
    else
        hipLaunchKernelGGL(sgd_tf_kernel<false>, dim3(grid_for(n >> 2, TPB)), dim3(TPB), 0, (hipStream_t)stream, param, grad,
                           (float*)nullptr, n, lr, 0.f, grad_scale, l2, skip, n_skip);
    return avsi_launch_status();
}
