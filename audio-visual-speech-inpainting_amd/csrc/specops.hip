// Stand-alone forms of the small DSP operators of audio_processing.py for callers that use them
// one at a time (the statistics tool, ASR-style front ends).  The inpainter itself goes through
// the fused front-end kernel (frontend.hip); these are plain memory-bound grid-stride kernels.
//   avsi_spectrogram_f32   |X|^power, optional log(. + eps)          (audio_processing.py:45-56)
//   avsi_logmel_f32        log(melW^T . spec + eps), band form        (audio_processing.py:59-72)
//   avsi_preemphasis_f32   y[t] = x[t] - alpha x[t-1], x[-1] = 0      (audio_processing.py:19-22)
//   avsi_delta_f32         regression deltas, cumulative SYMMETRIC pad (audio_processing.py:85-94)
#include "avsi_common.h"

namespace {
constexpr int TPB = 256;

inline int grid_for(int64_t items) {
    int64_t g = avsi_ceil_div(items, TPB);
    const int64_t cap = (int64_t)AVSI_NUM_CU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__global__ __launch_bounds__(TPB) void spectrogram_kernel(const float2* __restrict__ x, float* __restrict__ out,
                                                          int64_t n, float power, int do_log, float eps) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float2 v = x[i];
        float s = __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y);
        if (power != 1.f) s = (power == 2.f) ? s * s : __powf(s, power);
        if (do_log) s = __logf(s + eps);
        out[i] = s;
    }
}

__global__ __launch_bounds__(TPB) void logmel_kernel(const float* __restrict__ spec, int64_t ld, int64_t rows,
                                                     int num_mel, const int* __restrict__ start,
                                                     const int* __restrict__ len, const float* __restrict__ w,
                                                     int w_stride, float* __restrict__ out, float eps) {
    const int64_t items = rows * num_mel;
    for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < items; it += (int64_t)gridDim.x * TPB) {
        const int64_t r = it / num_mel;
        const int m = (int)(it - r * num_mel);
        const float* p = spec + r * ld + start[m];
        const float* wm = w + (int64_t)m * w_stride;
        float acc = 0.f;
        for (int j = 0; j < len[m]; ++j) acc = fmaf(p[j], wm[j], acc);
        out[it] = __logf(acc + eps);
    }
}

__global__ __launch_bounds__(TPB) void preemph_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t B,
                                                      int64_t N, int64_t ldx, float alpha) {
    const int64_t n = B * N;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const int64_t b = i / N, t = i - b * N;
        const float prev = t > 0 ? x[b * ldx + t - 1] : 0.f;
        y[i] = x[b * ldx + t] - alpha * prev;
    }
}

__global__ __launch_bounds__(TPB) void delta_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t B,
                                                    int T, int F, int N, float inv_denom) {
    const int64_t n = B * (int64_t)T * F;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const int f = (int)(i % F);
        const int64_t bt = i / F;
        const int t = (int)(bt % T);
        const float* row = x + (bt - t) * F + f;  // (b, 0, f)
        float acc = 0.f;
        for (int k = 1; k <= N; ++k) {
            const int hi = t + k < T ? t + k : T - 1, lo = t - k > 0 ? t - k : 0;  // repeated edge = SYMMETRIC pad 1, k times
            acc += (float)k * (row[(int64_t)hi * F] - row[(int64_t)lo * F]);
        }
        y[i] = acc * inv_denom;
    }
}
}  // namespace

extern "C" int avsi_spectrogram_f32(const float* stft, float* out, int64_t n, float power, int do_log, float eps,
                                    void* stream) {
    if (!stft || !out || n <= 0) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(spectrogram_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(stft), out, n, power, do_log, eps);
    return avsi_launch_status();
}

extern "C" int avsi_logmel_f32(const float* spec, int64_t ld, int64_t rows, int num_mel, const int32_t* mel_start,
                               const int32_t* mel_len, const float* mel_w, int mel_w_stride, float* out, float eps,
                               void* stream) {
    if (!spec || !out || !mel_start || !mel_len || !mel_w || rows <= 0 || num_mel <= 0) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(logmel_kernel, dim3(grid_for(rows * num_mel)), dim3(TPB), 0, (hipStream_t)stream, spec, ld, rows,
                       num_mel, mel_start, mel_len, mel_w, mel_w_stride, out, eps);
    return avsi_launch_status();
}

extern "C" int avsi_preemphasis_f32(const float* x, float* y, int64_t B, int64_t N, int64_t ldx, float alpha,
                                    void* stream) {
    if (!x || !y || B <= 0 || N <= 0 || ldx < N) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(preemph_kernel, dim3(grid_for(B * N)), dim3(TPB), 0, (hipStream_t)stream, x, y, B, N, ldx, alpha);
    return avsi_launch_status();
}

extern "C" int avsi_delta_f32(const float* x, float* y, int64_t B, int T, int F, int N, void* stream) {
    if (!x || !y || B <= 0 || T <= 0 || F <= 0 || N < 1) return AVSI_ERR_INVALID_ARG;
    int denom = 0;
    for (int k = 1; k <= N; ++k) denom += 2 * k * k;
    avsi_clear_error();
    hipLaunchKernelGGL(delta_kernel, dim3(grid_for(B * (int64_t)T * F)), dim3(TPB), 0, (hipStream_t)stream, x, y, B, T,
                       F, N, 1.f / (float)denom);
    return avsi_launch_status();
}
