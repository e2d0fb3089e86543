// LWS ("local weighted sums") phase reconstruction on gfx950 -- the phase refinement the reference's `infer`
// applies to every enhanced waveform when --oracle_phase is not given (inference.py:119,141-154, through the
// third-party `lws` package: lws.lws(384, 192, fftsize=512, mode='speech')).
//
// The package is neither vendored in the reference nor installable here; this file implements the published
// algorithm (Le Roux, Kameoka, Ono, Sagayama, DAFx-10 and ASJ 2010) under the conventions written down in
// oracle/lws.py -- the two are compared in tests/test_lws_gpu.py.  UNPINNED against the package itself.
//
// Four kernels:
//   lws_stft_kernel     sqrt-Hann (symmetric) analysis window zero-padded to the FFT length, 'perfectrec' padding
//   lws_stitch_kernel   the reference's mask stitching before / after the iterations (inference.py:143-153)
//   lws_sweeps_kernel   all "no future" / online / batch sweeps of one utterance, in place
//   lws_istft_*         inverse FFT with the perfect-reconstruction synthesis window, overlap-add, un-pad
//
// lws_sweeps_kernel is where the time goes.  A sweep visits the bins of the spectrogram in raster order IN PLACE:
// S[m,k] <- A[m,k] t/|t| with t the truncated consistency sum over rows m-1, m, m+1 and |p| <= L bins, for bins whose
// magnitude exceeds the sweep's threshold.  That order is strictly sequential along k inside a frame (the
// row-0 taps p < 0 were updated a moment ago) but only there: the taps of rows m-1, m+1 and the row-0 taps p > 0
// are known before the frame starts.  So a frame is two phases: (1) 64 lanes compute the 27 "known" taps of all
// 257 bins in parallel from an LDS ring of three spectrogram rows, (2) ONE lane per utterance runs the 5-tap
// recurrence over k with the last five updated bins in registers.  A workgroup is a single wave that owns U
// utterances (lane u < U runs the recurrence of utterance u), so there are no workgroup barriers, and thousands
// of such waves fill the chip: the recurrences are latency-bound, and independent waves on a SIMD hide each other.
#include <math.h>

#include "avsi_common.h"
#include "lws_shared.h"

namespace {

constexpr int NF = 512;          // FFT length = frame length of the (zero-padded) windows
constexpr int KB = NF / 2 + 1;   // 257 bins
constexpr int LMAX = 5;          // truncation of the consistency sum in frequency
constexpr int NP = 2 * LMAX + 1;
constexpr int RS = KB + 2 * LMAX + 1;  // complex entries per LDS row (mirror images either side; 268)
constexpr int PS = KB + 3;             // 260
constexpr int MAX_SWEEPS = 256;

constexpr int TAB_AWIN = 0, TAB_SWIN = NF, TAB_FLOATS = 2 * NF;

struct LwsWeights {       // alpha_q(p), q = -1, 0, +1, p = -L .. L   (re, im)
    float w[3][NP][2];
    int phase_step;       // 8 R / N ... the consistency phase factor is exp(-2 pi j (k + p) q R / N); R / N = phase_step / 64
};

struct LwsSchedule {
    int n;
    float rel[MAX_SWEEPS];            // threshold relative to the mean magnitude of the utterance
    unsigned char past_only[MAX_SWEEPS];
};

// ---------------------------------------------------------------------------------------------- windows (host, double)
void host_windows(int frame_len, int hop, int nfft, double* awin, double* swin) {
    // sqrt of the symmetric Hann window, synthesis window awin / sum_q awin(n + q R)^2, both zero-padded
    // symmetrically to nfft (oracle/lws.py LWS.__init__)
    const int lo = (nfft - frame_len) / 2;
    for (int i = 0; i < nfft; ++i) awin[i] = swin[i] = 0.0;
    const int Q = (frame_len + hop - 1) / hop;
    for (int i = 0; i < frame_len; ++i) awin[lo + i] = sqrt(0.5 * (1.0 - cos(2.0 * M_PI * i / (frame_len - 1))));
    for (int i = 0; i < frame_len; ++i) {
        double den = 0.0;
        for (int q = 0; q < Q; ++q) {
            const int j = i % hop + q * hop;
            if (j < frame_len) den += awin[lo + j] * awin[lo + j];
        }
        swin[lo + i] = awin[lo + i] / den;
    }
}

bool host_weights(int frame_len, int hop, int nfft, int L, LwsWeights& W) {
    double awin[NF], swin[NF];
    host_windows(frame_len, hop, nfft, awin, swin);
    // rows |q| >= 2 must vanish (the windows' supports do not overlap): frame_len <= 2 hop
    for (int q = -1; q <= 1; ++q)
        for (int p = -LMAX; p <= LMAX; ++p) {
            double re = 0.0, im = 0.0;
            if (p >= -L && p <= L)
                for (int n = 0; n < nfft; ++n) {
                    const int s = n - q * hop;
                    if (s < 0 || s >= nfft) continue;
                    const double pr = awin[n] * swin[s], ph = 2.0 * M_PI * p * n / nfft;
                    re += pr * cos(ph), im += pr * sin(ph);
                }
            W.w[q + 1][p + LMAX][0] = (float)(re / nfft), W.w[q + 1][p + LMAX][1] = (float)(im / nfft);
        }
    W.phase_step = 64 * hop / nfft;
    return true;
}

__global__ void lws_tables_kernel(float* tab, int frame_len, int hop, int nfft) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NF) return;
    const int lo = (nfft - frame_len) / 2;
    auto aw = [&](int j) -> double { return sqrt(0.5 * (1.0 - cospi(2.0 * j / (frame_len - 1)))); };
    double a = 0.0, s = 0.0;
    const int j = i - lo;
    if (i < nfft && j >= 0 && j < frame_len) {
        a = aw(j);
        double den = 0.0;
        for (int jj = j % hop; jj < frame_len; jj += hop) den += aw(jj) * aw(jj);
        s = a / den;
    }
    tab[TAB_AWIN + i] = (float)a;
    tab[TAB_SWIN + i] = (float)s;
}

// ---------------------------------------------------------------------------------------------- 512-point FFT in LDS
__device__ __forceinline__ int bitrev9(int x) { return (int)(__brev((unsigned)x) >> 23); }

// 512-point forward transform by ONE WAVE, eight values per lane: three radix-8 passes (Stockham: natural order in, natural
// order out), the 8-point transforms in registers, the data through LDS between the passes only (a radix-2 transform moves
// its 512 values through LDS nine times and was bound by LDS bandwidth: 24 GB for the frames of 1024 utterances in 0.6 ms).
// No block barrier: the LDS operations of a wave are performed in program order.  Lane j holds v[r] = x[j + 64 r] on entry;
// on return the spectrum sits in `s` at fft_slot(k).  tw[i] = exp(-2 pi j i / 512), i < 512.
constexpr int FFT_BUF = NF + NF / 8;                                          // 576 slots: one of padding behind every eight
__device__ __forceinline__ constexpr int fft_slot(int i) { return i + (i >> 3); }
__device__ __forceinline__ float2 cmulf(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
__device__ __forceinline__ float2 caddf(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csubf(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mulmi(float2 a) { return make_float2(a.y, -a.x); }       // a . (-i)

__device__ __forceinline__ void fft8(float2 (&v)[8]) {          // X[k] = sum_n v[n] exp(-2 pi j n k / 8), in place, natural order
    constexpr float R = 0.70710678118654752f;
    const float2 a0 = caddf(v[0], v[4]), a1 = csubf(v[0], v[4]), a2 = caddf(v[2], v[6]), a3 = mulmi(csubf(v[2], v[6]));
    const float2 a4 = caddf(v[1], v[5]), a5 = csubf(v[1], v[5]), a6 = caddf(v[3], v[7]), a7 = mulmi(csubf(v[3], v[7]));
    const float2 b0 = caddf(a0, a2), b2 = csubf(a0, a2), b1 = caddf(a1, a3), b3 = csubf(a1, a3);
    const float2 b4 = caddf(a4, a6), c6 = csubf(a4, a6), c5 = caddf(a5, a7), c7 = csubf(a5, a7);
    const float2 b5 = make_float2(R * (c5.x + c5.y), R * (c5.y - c5.x));      // . (1 - i) / sqrt 2
    const float2 b6 = mulmi(c6);                                              // . (-i)
    const float2 b7 = make_float2(R * (c7.y - c7.x), -R * (c7.x + c7.y));     // . (-1 - i) / sqrt 2
    v[0] = caddf(b0, b4), v[4] = csubf(b0, b4), v[1] = caddf(b1, b5), v[5] = csubf(b1, b5);
    v[2] = caddf(b2, b6), v[6] = csubf(b2, b6), v[3] = caddf(b3, b7), v[7] = csubf(b3, b7);
}

__device__ __forceinline__ void fft512_wave(float2 (&v)[8], float2* s, const float2* tw, int lane) {
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        const int Ns = pass == 0 ? 1 : (pass == 1 ? 8 : 64);
        const int k = lane & (Ns - 1);
        if (pass > 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = s[fft_slot(lane + 64 * r)];
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmulf(v[r], tw[r * k * (64 / Ns)]);
        }
        fft8(v);
        const int j0 = (lane / Ns) * Ns * 8 + k;
        asm volatile("" ::: "memory");          // (these writes stay behind the reads above, the next pass's reads behind them)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) s[fft_slot(j0 + r * Ns)] = v[r];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ void fill_twiddles(float2* tw, int tid) {      // 512 entries, 256 threads
#pragma unroll
    for (int i = tid; i < NF; i += 256) {
        float sn, cs;
        sincospif(-2.f * (float)i / 512.f, &sn, &cs);
        tw[i] = make_float2(cs, sn);
    }
}

constexpr int FPB = 4;        // frames per 256-thread block of the two transform kernels: one per wave

// spec [B][M][257] complex <- frames of the padded signal
__global__ __launch_bounds__(256) void lws_stft_kernel(const float* __restrict__ wav, int64_t wav_stride, int num_samples,
                                                       const float* __restrict__ tab, int hop, float2* __restrict__ spec,
                                                       int M) {
    __shared__ float2 sall[FPB][FFT_BUF], tw[NF];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = blockIdx.x * FPB + wv, b = blockIdx.y;
    fill_twiddles(tw, tid);
    __syncthreads();
    if (m >= M) return;
    float2* s = sall[wv];
    const int64_t t0 = (int64_t)m * hop - (NF - hop);       // 'perfectrec': N - R zeros in front
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        const int64_t t = t0 + n;
        const float x = (t >= 0 && t < num_samples) ? wav[(int64_t)b * wav_stride + t] : 0.f;
        v[r] = make_float2(x * tab[TAB_AWIN + n], 0.f);
    }
    fft512_wave(v, s, tw, lane);
    float2* out = spec + ((int64_t)b * M + m) * KB;
    for (int k = lane; k < KB; k += 64) out[k] = s[fft_slot(k)];
}

// frames [B][M][512] <- swin . irfft(spec)
__global__ __launch_bounds__(256) void lws_istft_frames_kernel(const float2* __restrict__ spec, const float* __restrict__ tab,
                                                               float* __restrict__ frames, int M) {
    __shared__ float2 sall[FPB][FFT_BUF], tw[NF];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = blockIdx.x * FPB + wv, b = blockIdx.y;
    fill_twiddles(tw, tid);
    __syncthreads();
    if (m >= M) return;
    float2* s = sall[wv];
    const float2* in = spec + ((int64_t)b * M + m) * KB;
    // x[n] = Re FFT(conj Xfull)[n] / N, Xfull[k] = X[k] (k <= 256), conj X[512 - k] above; irfft drops Im of DC / Nyquist
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        float2 x;
        if (k <= 256) {
            x = in[k];
            x.y = (k == 0 || k == 256) ? 0.f : -x.y;
        } else {
            x = in[NF - k];
        }
        v[r] = x;
    }
    fft512_wave(v, s, tw, lane);
    float* out = frames + ((int64_t)b * M + m) * NF;
#pragma unroll
    for (int i = 0; i < NF / 64; ++i) {
        const int n = lane + 64 * i;
        out[n] = s[fft_slot(n)].x * (1.f / NF) * tab[TAB_SWIN + n];
    }
}

__global__ void lws_ola_kernel(const float* __restrict__ frames, int M, int hop, float* __restrict__ out, int64_t out_stride,
                               int out_samples) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= out_samples) return;
    const int tp = t + (NF - hop);
    int m1 = tp / hop;
    int m0 = (tp - NF + hop) / hop;       // ceil((tp - N + 1) / R) for tp - N + 1 > 0
    if (tp - NF + 1 <= 0) m0 = 0;
    if (m1 > M - 1) m1 = M - 1;
    float acc = 0.f;
    for (int m = m0; m <= m1; ++m) acc += frames[((int64_t)b * M + m) * NF + (tp - m * hop)];
    out[(int64_t)b * out_stride + t] = acc;
}

// inference.py:143-153.  ref == null: "pre"  S <- |S| exp(j angle(S) mask_adj)
//                        ref != null: "post" S <- |S| exp(j (angle(ref) + angle(S) (1 - mask_adj)))
// mask_adj = mask on [0, mask_frames) x [0, mask_bins), zero elsewhere.
__global__ void lws_stitch_kernel(float2* __restrict__ spec, const float2* __restrict__ ref, const float* __restrict__ mask,
                                  int64_t msb, int64_t mst, int mask_frames, int mask_bins, int M, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % KB);
    const int m = (int)((i / KB) % M);
    const int64_t b = i / ((int64_t)KB * M);
    const float mk = (mask && m < mask_frames && k < mask_bins) ? mask[b * msb + (int64_t)m * mst + k] : 0.f;
    const float2 s = spec[i];
    const float mag = sqrtf(s.x * s.x + s.y * s.y);
    float ang;
    if (!ref) {
        if (mk == 1.f) return;                        // phase kept
        ang = atan2f(s.y, s.x) * mk;
    } else {
        const float2 r = ref[i];
        if (mk == 1.f) {                              // known phase restored: |S| e^{j angle(ref)}
            const float rm = sqrtf(r.x * r.x + r.y * r.y);
            spec[i] = rm > 0.f ? make_float2(mag * (r.x / rm), mag * (r.y / rm)) : make_float2(mag, 0.f);
            return;
        }
        ang = atan2f(r.y, r.x) + atan2f(s.y, s.x) * (1.f - mk);
    }
    float sn, cs;
    sincosf(ang, &sn, &cs);
    spec[i] = make_float2(mag * cs, mag * sn);
}

// ---------------------------------------------------------------------------------------------- the sweeps
__device__ __forceinline__ float2 cmadd(float2 acc, float2 w, float2 x) {
    return make_float2(fmaf(w.x, x.x, fmaf(-w.y, x.y, acc.x)), fmaf(w.x, x.y, fmaf(w.y, x.x, acc.y)));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// LDS of one wave (= one pipeline stage): its own ring of rows, partial sums and magnitudes
template <int U>
struct LwsWaveLds {
    float2 row[U][3][RS];   // ring of spectrogram rows m-1, m, m+1 with mirror images: entry j <-> bin j - LMAX
    float2 p[U][PS];        // phase-1 partial sums
    float amp[U][PS];       // magnitudes of row m
};

// v_rsq_f32 alone: rsqrtf() wraps it in a denormal-input rescue (scale, compare, two selects) that sits on the loop-carried
// chain of the in-frame recurrence; |t|^2 of a bin that is updated is nowhere near 1e-38
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }

// wave-local ordering point: LDS accesses of one wave execute in order, so lanes exchanging data through LDS only
// need the compiler not to reorder (and the counters drained before data another lane wrote is read)
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// rows are handed from one wave (sweep s - 1) to the next (sweep s) through global memory: device-scope accesses,
// i.e. never served from / parked in this CU's L1
__device__ __forceinline__ float2 row_load(const float2* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);      // one 8-byte access: a wave covers whole lines
    return make_float2(__builtin_bit_cast(float, (unsigned)v), __builtin_bit_cast(float, (unsigned)(v >> 32)));
}
__device__ __forceinline__ void row_store(float2* p, float2 v) {
    const unsigned long long w = (unsigned long long)__builtin_bit_cast(unsigned, v.x) |
                                 ((unsigned long long)__builtin_bit_cast(unsigned, v.y) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// NW waves per workgroup = NW sweeps of the same U utterances in flight, as a pipeline: wave w runs sweeps w, w + NW,
// ...; sweep s may work on row m once the sweep before it (the nearest one that touches anything) has finished row
// m + 1, the row it reads ahead -- exactly the raster-order dependence, so the result is bit-identical to running the
// sweeps one after the other.  (Row m + 2 is fetched one frame early when it is already final, otherwise after the frame.)
// Progress is a per-sweep row counter in LDS; rows travel through global memory (they stay in L2).  At small batches
// the sweeps of ONE utterance thus run on up to 16 waves instead of one (0.84 s -> ~60 ms per utterance); large
// batches use NW = 1 with several utterances per wave.
// mean and max magnitude of every utterance, from the untouched input (thresholds are relative to the mean; a sweep
// whose threshold is above the max touches nothing).  A kernel of its own so that every pipeline stage, whichever
// workgroup it sits in and whenever it starts, sees the same two numbers: stats[b] = (mean, max)
__global__ __launch_bounds__(256) void lws_stats_kernel(const float2* __restrict__ spec, int M, float2* __restrict__ stats) {
    __shared__ float s_sum[4], s_max[4];
    const float2* sp = spec + (int64_t)blockIdx.x * M * KB;
    float sum = 0.f, mx = 0.f;
    for (int i = threadIdx.x; i < M * KB; i += 256) {
        const float2 v = sp[i];
        const float a = sqrtf(v.x * v.x + v.y * v.y);
        sum += a, mx = fmaxf(mx, a);
    }
    sum = wave_sum(sum), mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum, s_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
        stats[blockIdx.x] = make_float2((s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]) / (float)(M * KB),
                                        fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])));
}

// G workgroups of NW waves per group of U utterances: G * NW pipeline stages.  Stages of one workgroup hand over
// through LDS counters, the last stage of a workgroup to the first stage of the next through a counter in global memory
// (`gdone`, one row of MAX_SWEEPS per utterance group, zero on entry).  All G workgroups of a group must be resident
// together (bounded waits, status word): the host keeps G * groups within the CU count.
template <int U, int NW>
__global__ __launch_bounds__(64 * NW) void lws_sweeps_kernel(float2* __restrict__ spec, int B, int M, const LwsWeights W,
                                                             const LwsSchedule sched, int* __restrict__ status,
                                                             const float2* __restrict__ stats, int* __restrict__ gdone_all,
                                                             int G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    volatile int* done = reinterpret_cast<volatile int*>(smem);                 // [MAX_SWEEPS] rows finished by sweep s
    const int wv = threadIdx.x >> 6;
    const int cluster = blockIdx.x / G, wg = blockIdx.x - cluster * G;
    int* gdone = gdone_all + (int64_t)cluster * MAX_SWEEPS;
    const int stage = wg * NW + wv, stages = G * NW;
    LwsWaveLds<U>& L = *reinterpret_cast<LwsWaveLds<U>*>(smem + MAX_SWEEPS * 4 + (size_t)wv * sizeof(LwsWaveLds<U>));
    auto& s_row = L.row;
    auto& s_p = L.p;
    auto& s_amp = L.amp;
    const int lane = threadIdx.x & 63;
    const int b0 = cluster * U;
    const int nu = min(U, B - b0);
    for (int i = threadIdx.x; i < MAX_SWEEPS; i += 64 * NW) done[i] = 0;
    __syncthreads();
    bool dead = false;       // a bounded wait gave up: stop waiting (results invalid, status word set), but finish

    // per-lane taps: bin k = lane + 64 i has (k + p) mod 64 = (lane + p) mod 64 for every i, and the phase factor
    // exp(-2 pi j (k + p) q R / N) has period 64 in k + p (host checks 64 R / N integer)
    float2 wm[NP], wp[NP], w0[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int kk = (lane + p - LMAX) & 63;
        float sn, cs;
        sincospif(2.f * (float)((kk * W.phase_step) & 63) / 64.f, &sn, &cs);       // e^{+2 pi j (k+p) R/N}: row q = -1
        const float2 am = make_float2(W.w[0][p][0], W.w[0][p][1]), ap = make_float2(W.w[2][p][0], W.w[2][p][1]);
        wm[p] = make_float2(am.x * cs - am.y * sn, am.x * sn + am.y * cs);
        wp[p] = make_float2(ap.x * cs + ap.y * sn, -ap.x * sn + ap.y * cs);         // conjugate factor: row q = +1
        w0[p] = make_float2(W.w[1][p][0], W.w[1][p][1]);
    }

    float mean_u[U], max_u[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float2 st = stats[min(b0 + u, B - 1)];
        mean_u[u] = st.x, max_u[u] = st.y;
    }
    __syncthreads();

    // wait until sweep s - 1 has finished `need` rows (bounded: see `dead`)
    auto wait_rows = [&](int pred, bool remote, int need) {
        if (pred < 0 || dead) return;
        need = need < M ? need : M;
        int spins = 0;
        while ((remote ? __hip_atomic_load(gdone + pred, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : done[pred]) < need) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 22)) {
                dead = true;
                if (lane == 0 && status) atomicOr(status, 1);
                break;
            }
        }
    };
    auto rows_done = [&](int pred, bool remote) -> int {      // wave-uniform; a sweep without predecessor sees all rows
        if (pred < 0 || dead) return 1 << 30;
        const int n = remote ? __hip_atomic_load(gdone + pred, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : done[pred];
        return __builtin_amdgcn_readfirstlane(n);
    };
    auto publish = [&](int s, int rows) {      // after every global store of those rows has completed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            done[s] = rows;
            if (G > 1) __hip_atomic_store(gdone + s, rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto load_row = [&](int u, int m, float2 (&r)[5]) {      // global -> registers (zeros outside the spectrogram)
        const bool ok = u < nu && m >= 0 && m < M;
        const float2* src = spec + ((int64_t)(b0 + u) * M + (ok ? m : 0)) * KB;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = lane + 64 * i;
            r[i] = (ok && k < KB) ? row_load(src + k) : make_float2(0.f, 0.f);
        }
    };
    auto store_row = [&](int u, int slot, const float2 (&r)[5]) {      // registers -> LDS row (bins only)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = lane + 64 * i;
            if (k < KB) s_row[u][slot][k + LMAX] = r[i];
        }
    };
    auto mirror_row = [&](int u, int slot) {      // mirror images below DC and above Nyquist
        if (lane >= 1 && lane <= LMAX) {
            const float2 lo = s_row[u][slot][LMAX + lane], hi = s_row[u][slot][LMAX + 256 - lane];
            s_row[u][slot][LMAX - lane] = make_float2(lo.x, -lo.y);
            s_row[u][slot][LMAX + 256 + lane] = make_float2(hi.x, -hi.y);
        }
    };

    // The sweeps that touch anything ("active": some utterance of the group has a magnitude above the threshold) are
    // dealt to the stages in turn -- the a-th active sweep runs on stage a % stages -- and each follows the active sweep
    // before it.  An idle sweep leaves the rows as they were: it needs no stage (dealing by sweep NUMBER gave the stages
    // of the 7 idle sweeps of the 'speech' schedule one sweep less and the others 7 instead of 6 at 16 stages), and
    // waiting for it would only serialise the pipeline (it could report a row finished no earlier than its own
    // predecessor's LAST row).  Every stage of the group sees the same thresholds and statistics, hence the same deal.
    int rank_a = -1, last_active = -1;
    for (int sw = 0; sw < sched.n; ++sw) {
        float thr[U];
        bool any_u = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            thr[u] = sched.rel[sw] * mean_u[u];
            any_u |= (u < nu) && (max_u[u] > thr[u]);
        }
        if (!any_u) continue;       // wave-uniform
        const int pred = last_active;
        last_active = sw, ++rank_a;
        if (rank_a % stages != stage) continue;
        const bool pred_remote = pred >= 0 && G > 1 && ((rank_a - 1) % stages) / NW != wg;     // its stage sits in another workgroup
        const bool past_only = sched.past_only[sw] != 0;

        // ring: slot (m + 3) % 3 holds row m
        wait_rows(pred, pred_remote, 2);
        wave_sync();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float2 r[5];
            load_row(u, -1, r);
            store_row(u, 2, r);
            load_row(u, 0, r);
            store_row(u, 0, r);
            load_row(u, 1, r);
            store_row(u, 1, r);
        }
        wave_sync();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mirror_row(u, 0);
            mirror_row(u, 1);
            mirror_row(u, 2);
        }
        wave_sync();

        for (int m = 0; m < M; ++m) {
            const int sc = m % 3, sp_ = (m + 2) % 3, sn_ = (m + 1) % 3;   // cur, prev, next
            // Row m + 2 (it lands in the slot of row m - 1 after this frame) is final once sweep `pred` is past it.  If it
            // already is, the row is requested now and its latency hides behind this frame; if not, the frame does NOT
            // wait for it -- everything it reads (rows m and m + 1) is here -- and fetches the row after its own row has
            // been published: a sweep then starts two rows behind its predecessor instead of three, which with ~100
            // sweeps in a chain is most of what one utterance waits for.
            const bool early = rows_done(pred, pred_remote) >= (m + 3 < M ? m + 3 : M);
            float2 pre[U][5];
            if (early) {
#pragma unroll
                for (int u = 0; u < U; ++u) load_row(u, m + 2, pre[u]);
            }

            // magnitudes of row m, and which bins are above the threshold for at least one utterance of the wave?  (one
            // bit per bin: the magnitudes never change -- an update only turns a bin -- so a bin that is below now is
            // copied, not computed, by the recurrence below)
            unsigned long long amask[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int k = lane + 64 * i;
                bool act = false;
                if (k < KB) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float2 v = s_row[u][sc][k + LMAX];
                        const float a = sqrtf(v.x * v.x + v.y * v.y);
                        s_amp[u][k] = a;
                        act |= (u < nu) && (a > thr[u]);
                    }
                }
                amask[i] = __ballot(act);
            }
            const bool frame_active = (amask[0] | amask[1] | amask[2] | amask[3] | amask[4]) != 0ull;

            if (frame_active) {
                // ---- phase 1: the taps that are known before the frame starts
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const int k = lane + 64 * i;
                        if (amask[i] == 0ull) continue;                 // 64 bins none of which is updated in this sweep
                        if (k < KB) {
                            float2 acc = make_float2(0.f, 0.f);
                            const float2* rp = &s_row[u][sp_][k];        // entry k + p + LMAX, p = -LMAX ..
                            const float2* rn = &s_row[u][sn_][k];
                            const float2* rc = &s_row[u][sc][k];
#pragma unroll
                            for (int p = 0; p < NP; ++p) acc = cmadd(acc, wm[p], rp[p]);
                            if (!past_only) {
#pragma unroll
                                for (int p = 0; p < NP; ++p) acc = cmadd(acc, wp[p], rn[p]);
#pragma unroll
                                for (int p = LMAX + 1; p < NP; ++p)
                                    if (k + p - LMAX <= 256) acc = cmadd(acc, w0[p], rc[p]);
                            }
                            s_p[u][k] = acc;
                        }
                    }
                }
                wave_sync();
                if (past_only) {
                    // "no future" pass: rows q < 0 only, no dependence inside the frame
#pragma unroll
                    for (int u = 0; u < U; ++u) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            const int k = lane + 64 * i;
                            if (k < KB) {
                                const float2 t = s_p[u][k];
                                const float a = s_amp[u][k], n2 = t.x * t.x + t.y * t.y;
                                if (u < nu && a > thr[u] && n2 > 0.f) {
                                    const float sc_ = a * fast_rsqrt(n2);
                                    s_row[u][sc][k + LMAX] = make_float2(t.x * sc_, t.y * sc_);
                                }
                            }
                        }
                    }
                } else if (lane < nu) {
                    // ---- phase 2: the in-frame recurrence, one lane per utterance
                    const int u = lane;
                    float2* row = s_row[u][sc];
                    const float th = thr[0 + 0 * u];       // thr[] is lane-uniform per u: select below
                    float thu = th;
#pragma unroll
                    for (int uu = 1; uu < U; ++uu) thu = (u == uu) ? thr[uu] : thu;
                    const float2 c1 = make_float2(W.w[1][LMAX - 1][0], W.w[1][LMAX - 1][1]);
                    const float2 c2 = make_float2(W.w[1][LMAX - 2][0], W.w[1][LMAX - 2][1]);
                    const float2 c3 = make_float2(W.w[1][LMAX - 3][0], W.w[1][LMAX - 3][1]);
                    const float2 c4 = make_float2(W.w[1][LMAX - 4][0], W.w[1][LMAX - 4][1]);
                    const float2 c5 = make_float2(W.w[1][LMAX - 5][0], W.w[1][LMAX - 5][1]);
                    const float2 cpast[LMAX] = {c1, c2, c3, c4, c5};
                    // Lower edge (bins 0 .. LMAX-1): their taps below bin 0 are mirror images of bins 1 .. LMAX, which this
                    // same sweep updates -- the 2*LMAX values involved are held in registers (the loop is fully unrolled) and
                    // every change goes to the row AND to its image, so no bin waits for an LDS round trip.
                    float2 e[2 * LMAX];
#pragma unroll
                    for (int i = 0; i < 2 * LMAX; ++i) e[i] = row[i];           // bins -LMAX .. LMAX-1
#pragma unroll
                    for (int k = 0; k < LMAX; ++k) {
                        const float a = s_amp[u][k];
                        float2 t = s_p[u][k];
#pragma unroll
                        for (int p = LMAX; p >= 1; --p) t = cmadd(t, cpast[p - 1], e[k - p + LMAX]);   // newest tap last
                        const float n2 = t.x * t.x + t.y * t.y;
                        const bool upd = (a > thu) && (n2 > 0.f);
                        const float sc_ = a * fast_rsqrt(n2);
                        const float2 v = upd ? make_float2(t.x * sc_, t.y * sc_) : e[k + LMAX];
                        e[k + LMAX] = v;
                        row[k + LMAX] = v;
                        if (k >= 1) {
                            const float2 im = upd ? make_float2(v.x, -v.y) : e[LMAX - k];
                            e[LMAX - k] = im;
                            row[LMAX - k] = im;
                        }
                    }
                    float2 s1 = e[2 * LMAX - 1], s2 = e[2 * LMAX - 2], s3 = e[2 * LMAX - 3], s4 = e[2 * LMAX - 4],
                           s5 = e[2 * LMAX - 5];
                    // Software-pipelined straight-line body. Only nu <= 4 lanes run here, so the latency of the loop-carried
                    // chain is fully exposed: it is kept to the NEWEST tap (one complex MAC), the norm, a reciprocal
                    // square root and two multiplies, and the four older taps of the NEXT bin (which need nothing of the
                    // bin in flight) sit beside it in the same basic block to fill its issue slots. The store is
                    // unconditional (an inactive bin writes its old value back); LDS operands are requested two bins ahead.
                    float2 p1 = s_p[u][LMAX + 1], o0 = row[LMAX + LMAX], o1 = row[LMAX + 1 + LMAX];
                    float a0 = s_amp[u][LMAX], a1 = s_amp[u][LMAX + 1];
                    float2 part = cmadd(cmadd(cmadd(cmadd(s_p[u][LMAX], c5, s5), c4, s4), c3, s3), c2, s2);
#pragma unroll 4
                    for (int k = LMAX; k <= 256 - LMAX - 1; ++k) {
                        const float2 old = o0;
                        const float a = a0;
                        const float2 p2 = s_p[u][k + 2], o2 = row[k + 2 + LMAX];    // <= 258: inside the padded rows
                        const float a2 = s_amp[u][k + 2];
                        const float2 t = cmadd(part, c1, s1);
                        float2 nxt = cmadd(p1, c5, s4);          // bin k + 1 without its newest tap
                        nxt = cmadd(nxt, c4, s3);
                        nxt = cmadd(nxt, c3, s2);
                        nxt = cmadd(nxt, c2, s1);
                        const float n2 = t.x * t.x + t.y * t.y;
                        const bool upd = (a > thu) && (n2 > 0.f);
                        const float sc_ = a * fast_rsqrt(n2);
                        const float2 v = upd ? make_float2(t.x * sc_, t.y * sc_) : old;
                        row[k + LMAX] = v;
                        s4 = s3, s3 = s2, s2 = s1, s1 = v;
                        part = nxt, p1 = p2, o0 = o1, o1 = o2, a0 = a1, a1 = a2;
                    }
                    // Upper edge (bins 256-LMAX .. 256): besides the LMAX past bins, their taps above bin 256 are mirror images
                    // of bins 256-LMAX .. 255, updated by this same block -- registers again, newest tap last.
                    float2 q[2 * LMAX + 1], img[LMAX + 1];
                    q[0] = row[256 - 2 * LMAX + LMAX], q[1] = s4, q[2] = s3, q[3] = s2, q[4] = s1;   // bins 256-2*LMAX .. 256-LMAX-1
                    static_assert(LMAX == 5, "the past-tap registers s1..s4 above are written out for LMAX = 5");
#pragma unroll
                    for (int p = 1; p <= LMAX; ++p) img[p] = row[256 + p + LMAX];
#pragma unroll
                    for (int j = 0; j <= LMAX; ++j) {
                        const int k = 256 - LMAX + j;
                        const float a = s_amp[u][k];
                        const float2 old = row[k + LMAX];
                        float2 t = s_p[u][k];
#pragma unroll
                        for (int p = LMAX; p >= 1; --p)
                            if (k + p > 256) t = cmadd(t, make_float2(W.w[1][LMAX + p][0], W.w[1][LMAX + p][1]), img[k + p - 256]);
#pragma unroll
                        for (int p = LMAX; p >= 1; --p) t = cmadd(t, cpast[p - 1], q[j + LMAX - p]);
                        const float n2 = t.x * t.x + t.y * t.y;
                        const bool upd = (a > thu) && (n2 > 0.f);
                        const float sc_ = a * fast_rsqrt(n2);
                        const float2 v = upd ? make_float2(t.x * sc_, t.y * sc_) : old;
                        q[j + LMAX] = v;
                        row[k + LMAX] = v;
                        if (k <= 255) {
                            const float2 im = upd ? make_float2(v.x, -v.y) : img[256 - k];
                            img[256 - k] = im;
                            row[LMAX + 512 - k] = im;
                        }
                    }
                }
                wave_sync();
                // refresh the mirror images of row m (it becomes row m - 1 of the next frame) and write it back
#pragma unroll
                for (int u = 0; u < U; ++u) mirror_row(u, sc);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u < nu) {
                        float2* dst = spec + ((int64_t)(b0 + u) * M + m) * KB;
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            const int k = lane + 64 * i;
                            if (k < KB) row_store(dst + k, s_row[u][sc][k + LMAX]);
                        }
                    }
                }
            }
            publish(sw, m + 1);
            if (!early) {
                wait_rows(pred, pred_remote, m + 3);
#pragma unroll
                for (int u = 0; u < U; ++u) load_row(u, m + 2, pre[u]);
            }
            wave_sync();
            // row m + 2 replaces row m - 1
#pragma unroll
            for (int u = 0; u < U; ++u) store_row(u, sp_, pre[u]);
            wave_sync();
#pragma unroll
            for (int u = 0; u < U; ++u) mirror_row(u, sp_);
            wave_sync();
        }
    }
}

bool geometry_ok(int frame_len, int hop, int nfft) {
    return nfft == NF && frame_len >= 4 && frame_len <= nfft && hop > 0 && hop <= frame_len && frame_len <= 2 * hop &&
           ((nfft - frame_len) / 2 + frame_len <= nfft) && (64 * hop) % nfft == 0;
}

}  // namespace

// ---- shared with lws_skew.hip (lws_shared.h)
void avsi_lws_host_alpha(int frame_len, int hop, int nfft, int L, double alpha[3][AVSI_LWS_NP][2]) {
    double awin[NF], swin[NF];
    host_windows(frame_len, hop, nfft, awin, swin);
    for (int q = -1; q <= 1; ++q)
        for (int p = -LMAX; p <= LMAX; ++p) {
            double re = 0.0, im = 0.0;
            if (p >= -L && p <= L)
                for (int n = 0; n < nfft; ++n) {
                    const int s = n - q * hop;
                    if (s < 0 || s >= nfft) continue;
                    const double pr = awin[n] * swin[s], ph = 2.0 * M_PI * p * n / nfft;
                    re += pr * cos(ph), im += pr * sin(ph);
                }
            alpha[q + 1][p + LMAX][0] = re / nfft, alpha[q + 1][p + LMAX][1] = im / nfft;
        }
}
bool avsi_lws_geometry_ok(int frame_len, int hop, int nfft) { return geometry_ok(frame_len, hop, nfft); }
bool avsi_lws_make_schedule(int nofuture_iterations, float nofuture_alpha, int online_iterations, float online_alpha,
                            int batch_iterations, float batch_alpha, float batch_beta, float batch_gamma, AvsiLwsSchedule& S) {
    if (nofuture_iterations + online_iterations + batch_iterations > AVSI_LWS_MAX_SWEEPS) return false;
    S.n = 0;
    for (int i = 0; i < nofuture_iterations; ++i) S.rel[S.n] = nofuture_alpha, S.past_only[S.n++] = 1;
    for (int i = 0; i < online_iterations; ++i) S.rel[S.n] = online_alpha, S.past_only[S.n++] = 0;
    for (int i = 0; i < batch_iterations; ++i)
        S.rel[S.n] = (float)(batch_alpha * exp(-(double)batch_beta * pow((double)i, (double)batch_gamma))), S.past_only[S.n++] = 0;
    for (int i = S.n; i < AVSI_LWS_MAX_SWEEPS; ++i) S.rel[i] = 0.f, S.past_only[i] = 0;
    return true;
}
void avsi_lws_launch_stats(const float* spec, int batch, int M, float* stats, hipStream_t st) {
    hipLaunchKernelGGL(lws_stats_kernel, dim3(batch), dim3(256), 0, st, reinterpret_cast<const float2*>(spec), M,
                       reinterpret_cast<float2*>(stats));
}

extern "C" int avsi_lws_num_frames(int num_samples, int hop, int nfft) {
    if (num_samples <= 0 || hop <= 0 || nfft < hop) return 0;
    const int64_t padded = (int64_t)num_samples + 2 * (int64_t)(nfft - hop);
    const int64_t m = (padded - nfft + hop - 1) / hop + 1;
    return m < 1 ? 1 : (int)m;
}

extern "C" size_t avsi_lws_table_floats(int frame_len, int hop, int nfft) {
    return geometry_ok(frame_len, hop, nfft) ? (size_t)TAB_FLOATS : 0;
}

extern "C" int avsi_lws_init_tables(float* table, int frame_len, int hop, int nfft, void* stream) {
    if (!table) return AVSI_ERR_INVALID_ARG;
    if (!geometry_ok(frame_len, hop, nfft)) return AVSI_ERR_UNSUPPORTED;
    avsi_clear_error();
    hipLaunchKernelGGL(lws_tables_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, table, frame_len, hop, nfft);
    return avsi_launch_status();
}

extern "C" int avsi_lws_stft_f32(const float* wav, int64_t wav_stride, int batch, int num_samples, const float* table, int hop,
                                 int nfft, float* spec, int num_frames, void* stream) {
    if (!wav || !table || !spec || batch <= 0 || num_samples <= 0 || (batch > 1 && wav_stride < num_samples))
        return AVSI_ERR_INVALID_ARG;
    if (nfft != NF || hop <= 0 || hop > nfft) return AVSI_ERR_UNSUPPORTED;
    if (num_frames != avsi_lws_num_frames(num_samples, hop, nfft) || batch > 65535) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(lws_stft_kernel, dim3((num_frames + FPB - 1) / FPB, batch), dim3(256), 0, (hipStream_t)stream, wav, wav_stride, num_samples,
                       table, hop, reinterpret_cast<float2*>(spec), num_frames);
    return avsi_launch_status();
}

extern "C" int avsi_lws_stitch_f32(float* spec, const float* ref, const float* mask, int64_t mask_stride_b,
                                   int64_t mask_stride_t, int mask_frames, int mask_bins, int batch, int num_frames, int nfft,
                                   void* stream) {
    if (!spec || batch <= 0 || num_frames <= 0 || (mask && (mask_frames < 0 || mask_bins < 0))) return AVSI_ERR_INVALID_ARG;
    if (nfft != NF) return AVSI_ERR_UNSUPPORTED;
    const int64_t total = (int64_t)batch * num_frames * KB;
    avsi_clear_error();
    hipLaunchKernelGGL(lws_stitch_kernel, dim3((unsigned)avsi_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float2*>(spec), reinterpret_cast<const float2*>(ref), mask, mask_stride_b, mask_stride_t,
                       mask_frames, mask_bins, num_frames, total);
    return avsi_launch_status();
}

// word 0: status; then (mean, max) per utterance; then one row of progress counters per utterance group
extern "C" size_t avsi_lws_run_workspace_bytes(int batch) {
    return batch > 0 ? 16 + (size_t)batch * sizeof(float2) + (size_t)batch * MAX_SWEEPS * sizeof(int) : 0;
}

extern "C" int avsi_lws_run_f32(float* spec, int batch, int num_frames, int frame_len, int hop, int nfft, int L,
                                int nofuture_iterations, float nofuture_alpha, int online_iterations, float online_alpha,
                                int batch_iterations, float batch_alpha, float batch_beta, float batch_gamma,
                                int utterances_per_wave, int waves_per_group, int groups_per_utterance, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (!spec || batch <= 0 || num_frames <= 0 || L < 1 || nofuture_iterations < 0 || online_iterations < 0 ||
        batch_iterations < 0)
        return AVSI_ERR_INVALID_ARG;
    if (!geometry_ok(frame_len, hop, nfft) || L > LMAX ||
        nofuture_iterations + online_iterations + batch_iterations > MAX_SWEEPS)
        return AVSI_ERR_UNSUPPORTED;
    LwsWeights W;
    host_weights(frame_len, hop, nfft, L, W);
    LwsSchedule S;
    S.n = 0;
    for (int i = 0; i < nofuture_iterations; ++i) S.rel[S.n] = nofuture_alpha, S.past_only[S.n++] = 1;
    for (int i = 0; i < online_iterations; ++i) S.rel[S.n] = online_alpha, S.past_only[S.n++] = 0;
    for (int i = 0; i < batch_iterations; ++i)
        S.rel[S.n] = (float)(batch_alpha * exp(-(double)batch_beta * pow((double)i, (double)batch_gamma))), S.past_only[S.n++] = 0;
    for (int i = S.n; i < MAX_SWEEPS; ++i) S.rel[i] = 0.f, S.past_only[i] = 0;
    if (S.n == 0) return AVSI_OK;
    // Shape of the launch: U utterances per wave (lane u < U runs the recurrence of utterance u), NW waves per
    // workgroup and G workgroups per utterance group (G * NW sweeps of those utterances in flight as a pipeline).
    // Small batches want their sweeps spread over waves and CUs (latency: 0.84 s per utterance on one wave); large
    // batches fill the chip with utterances instead.
    int U = utterances_per_wave, NW = waves_per_group;
    // Launch shape.  One lane per utterance runs the in-frame recurrence, so a wave should carry several utterances
    // (U, at most 4 next to a useful number of waves: a CU's LDS holds 16 utterance-waves); the sweeps of an utterance
    // group are pipelined over the NW waves of a workgroup and over G workgroups, which must all be resident.  The
    // recurrence is issue-bound per wave, so waves that share a SIMD slow each other down: spread over CUs first.
    // Measured (tools/lws_shapes.sh, ms per batch, U x NW x G): 8 utterances 1x4x26: 23.6, 1x8x13: 28.7; 32: 1x8x8: 23.4,
    // 4x4x16: 23.7, 2x8x8: 25.0, 1x4x8: 26.9; 64: 4x4x16: 22.0, 2x8x8: 24.9, 1x8x4: 31.1; 100: 4x4x10: 27.3, 2x8x5: 30.5,
    // 1x8x2: 50.0; 256: 4x4x4: 47.0, 2x8x2: 54.1, 1x8x1: 90.0; 512: 4x4x2: 83.9, 2x8x1: 98.3; 1024: 4x4x1: 6.5 k utterances/s.
    // 17 .. 32 utterances: two per wave, 8 waves, so that 16 utterance groups x 13 workgroups give every sweep a stage of
    // its own (32 utterances: 2x8x16 14.8 ms, 1x8x8 17.5, 4x4x16 17.1 -- since the chain of sweeps got shorter, section 4.3d
    // of DESIGN.md, more stages pay even with two waves to a SIMD)
    if (U == 0) U = batch > 32 ? 4 : batch > 16 ? 2 : 1;
    const int clusters = (batch + U - 1) / U;
    if (NW == 0) {
        if (U > 1) NW = 16 / U;
        else if (batch > AVSI_NUM_CU) NW = 8;
        else {
            // the fewest waves per workgroup with which the utterance still gets ~64 pipeline stages
            const int gmax = AVSI_NUM_CU / clusters;
            NW = 16;
            for (int nw = 4; nw <= 8; nw *= 2) {
                const int useful = (S.n + nw - 1) / nw;
                if ((gmax < useful ? gmax : useful) * nw >= 64) {
                    NW = nw;
                    break;
                }
            }
        }
    }
    int G = groups_per_utterance;
    if (G == 0) {      // as many workgroups per utterance group as the chip holds, at most one per NW sweeps
        G = clusters <= AVSI_NUM_CU ? AVSI_NUM_CU / clusters : 1;
        const int useful = (S.n + NW - 1) / NW;
        G = G < 1 ? 1 : (G > useful ? useful : G);
    }
    if (G < 1 || (G > 1 && (int64_t)G * clusters > AVSI_NUM_CU)) return AVSI_ERR_INVALID_ARG;     // all must be resident
    if (!workspace || workspace_bytes < avsi_lws_run_workspace_bytes(batch)) return AVSI_ERR_WORKSPACE;
    const hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, avsi_lws_run_workspace_bytes(batch), st) != hipSuccess) return AVSI_ERR_LAUNCH;
    int* status = static_cast<int*>(workspace);
    float2* stats = reinterpret_cast<float2*>(static_cast<char*>(workspace) + 16);
    int* gdone = reinterpret_cast<int*>(static_cast<char*>(workspace) + 16 + (size_t)batch * sizeof(float2));
    avsi_clear_error();
    float2* sp = reinterpret_cast<float2*>(spec);
    hipLaunchKernelGGL(lws_stats_kernel, dim3(batch), dim3(256), 0, st, sp, num_frames, stats);
#define AVSI_LWS_LAUNCH(UV, NWV)                                                                                          \
    do {                                                                                                                   \
        const size_t lds = MAX_SWEEPS * 4 + (size_t)(NWV) * sizeof(LwsWaveLds<UV>);                                        \
        (void)hipFuncSetAttribute((const void*)lws_sweeps_kernel<UV, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                  (int)lds);                                                                               \
        hipLaunchKernelGGL((lws_sweeps_kernel<UV, NWV>), dim3(clusters * G), dim3(64 * (NWV)), lds, st, sp, batch, num_frames, W, \
                           S, status, stats, gdone, G);                                                                    \
    } while (0)
    if (U == 4 && NW == 1) AVSI_LWS_LAUNCH(4, 1);
    else if (U == 2 && NW == 1) AVSI_LWS_LAUNCH(2, 1);
    else if (U == 1 && NW == 1) AVSI_LWS_LAUNCH(1, 1);
    else if (U == 1 && NW == 4) AVSI_LWS_LAUNCH(1, 4);
    else if (U == 1 && NW == 8) AVSI_LWS_LAUNCH(1, 8);
    else if (U == 1 && NW == 16) AVSI_LWS_LAUNCH(1, 16);
    else if (U == 2 && NW == 8) AVSI_LWS_LAUNCH(2, 8);
    else if (U == 4 && NW == 4) AVSI_LWS_LAUNCH(4, 4);
    else return AVSI_ERR_INVALID_ARG;
#undef AVSI_LWS_LAUNCH
    return avsi_launch_status();
}

extern "C" size_t avsi_lws_istft_workspace_bytes(int batch, int num_frames, int nfft) {
    if (batch <= 0 || num_frames <= 0 || nfft != NF) return 0;
    return (size_t)batch * num_frames * NF * sizeof(float);
}

extern "C" int avsi_lws_istft_f32(const float* spec, int batch, int num_frames, const float* table, int hop, int nfft,
                                  float* out, int64_t out_stride, int out_samples, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    if (!spec || !table || !out || batch <= 0 || num_frames <= 0 || out_samples <= 0 || batch > 65535) return AVSI_ERR_INVALID_ARG;
    if (nfft != NF || hop <= 0 || hop > nfft) return AVSI_ERR_UNSUPPORTED;
    const int64_t avail = (int64_t)(num_frames - 1) * hop + nfft - 2 * (int64_t)(nfft - hop);
    if (out_samples > avail || (batch > 1 && out_stride < out_samples)) return AVSI_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < avsi_lws_istft_workspace_bytes(batch, num_frames, nfft)) return AVSI_ERR_WORKSPACE;
    avsi_clear_error();
    const hipStream_t st = (hipStream_t)stream;
    float* frames = static_cast<float*>(workspace);
    hipLaunchKernelGGL(lws_istft_frames_kernel, dim3((num_frames + FPB - 1) / FPB, batch), dim3(256), 0, st, reinterpret_cast<const float2*>(spec),
                       table, frames, num_frames);
    hipLaunchKernelGGL(lws_ola_kernel, dim3((unsigned)avsi_ceil_div(out_samples, 256), batch), dim3(256), 0, st, frames, num_frames,
                       hop, out, out_stride, out_samples);
    return avsi_launch_status();
}
