// Waveform reconstruction on gfx950: (de-normalise + phase) -> inverse real FFT -> synthesis window
// -> overlap-add, one pass over HBM.
//
// Replaces tf.contrib.signal.inverse_stft with inverse_stft_window_fn (reference
// audio_processing.py:145-157) and, in its fused mode, the ops in front of it in
// StackedBLSTMModel.enhanced_sources (models.py:181-197):
//     mag = exp(pred * std + mean);  phi = angle(target_stft * mask)   [mask = 1: oracle phase]
//     X   = mag (cos phi + j sin phi)                                  (audio_processing.py:160-164)
// cos/sin(angle(S)) are Re S / |S| and Im S / |S| (angle(0) = 0 => X = mag), so no atan2 / sincos
// is evaluated.  The generic modes take a complex spectrogram or (magnitude, phase) planes.
// Mode 3 is mode 2 WITHOUT the complex target spectrogram: the tile's 16 frames of the target WAVEFORM (13 KB instead
// of 33 KB of complex bins, which the front end would have had to write first) are transformed forward in the kernel --
// the front end's own 16 x 16 FFT and arithmetic (frontend.hip) -- and only their phase is used.
//
// Tile = 15 output hops of one utterance: the workgroup inverse-transforms the 16 frames that touch
// them (one halo frame) with the same 16 x 16 in-register FFT as the front end
// (z = conj(FFT256(conj Z)) / 256, Z[k] = E[k] + j O[k] built from X[k] and X[256 - k]), applies
// w[n] / sum_k w^2 and adds the two frames that cover each sample in LDS.
// Every output sample is written exactly once with coalesced stores; no atomics.
#include "avsi_common.h"
#include "fft16.h"

using namespace avsi_fft;

namespace {

constexpr int FR = 16;
constexpr int TPB = 256;
constexpr int ZSTRIDE = 272;
constexpr int XS = 258;  // complex per frame in the X tile

constexpr int TAB_WIN = 0;  // synthesis window [512] (zero past frame_len)
constexpr int TAB_TW256 = 512;
constexpr int TAB_TW512 = 1024;
constexpr int TAB_AWIN = 512 + 512 + 2 * 257 + 2;      // analysis window [512] (periodic Hann, zero past frame_len): mode 3
constexpr int TAB_FLOATS = TAB_AWIN + 512;

__global__ void istft_tables_kernel(float* tab, int frame_len, int hop) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 512) {
        float v = 0.f;
        if (i < frame_len) {
            // inverse_stft_window_fn: w[n] / sum_k w[(n mod hop) + k hop]^2, periodic Hann forward window
            double den = 0.0;
            for (int j = i % hop; j < frame_len; j += hop) {
                const double wj = 0.5 - 0.5 * cospi(2.0 * j / frame_len);
                den += wj * wj;
            }
            v = (float)((0.5 - 0.5 * cospi(2.0 * i / frame_len)) / den);
        }
        tab[TAB_WIN + i] = v;
        tab[TAB_AWIN + i] = i < frame_len ? (float)(0.5 - 0.5 * cospi(2.0 * i / frame_len)) : 0.f;     // frontend.hip's window
    }
    if (i < 256) {
        tab[TAB_TW256 + 2 * i] = (float)cospi(2.0 * i / 256.0);
        tab[TAB_TW256 + 2 * i + 1] = (float)(-sinpi(2.0 * i / 256.0));
    }
    if (i <= 256) {
        tab[TAB_TW512 + 2 * i] = (float)cospi(2.0 * i / 512.0);
        tab[TAB_TW512 + 2 * i + 1] = (float)(-sinpi(2.0 * i / 512.0));
    }
}

// MODE / HAS_MASK are compile-time so that the 16 frames' loads of a tile are straight-line code: with the mode
// tested at run time each frame's loads sat in their own blocks, closed by s_waitcnt vmcnt(0) -- sixteen memory
// latencies in a row per tile.
// Built for three waves per SIMD (<= 168 registers; mode 3 spills ~30 of its 197 to scratch and still gains: 2.30 -> 1.99 ms at
// 4096 utterances, measured against the two-wave build on one box).  LATE (mode 3): the prediction / mask loads of a tile are
// issued behind the forward transform instead of in front of it (the other waves cover their latency: 2.03 -> 1.99 ms).
template <int MODE, bool HAS_MASK, bool LATE = (MODE == 3)>
__global__ __launch_bounds__(TPB, 3) void istft_kernel(const avsi_istft_args a, const int tiles_per_utt, const int n_tiles,
                                                    const int n_hops, const int step) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // s_x (spectrum tile) and s_z (FFT transposition scratch) share storage: s_x is dead once every lane has
    // gathered its 16 inputs of the first FFT stage (barrier below).  With frames kept at their true length this
    // is 59 KB per workgroup instead of 100 KB: two workgroups per CU.
    cf* s_x = reinterpret_cast<cf*>(smem);                             // [FR][XS]   spectrum tile
    cf* s_z = s_x;                                                     // [FR][ZSTRIDE] FFT scratch (aliases s_x)
    // round 5: the windowed frames share that storage too -- they are written from registers after the last read of s_z
    // (the barrier in front of the second FFT) and are dead before the next tile writes s_x / s_z (the barrier behind the
    // overlap-add).  35 KB per workgroup (48 KB in mode 3, whose samples need a place of their own while s_z is written)
    // instead of 59: THREE workgroups per CU, and the kernel is bound by vector-instruction issue at two waves per SIMD.
    float* s_f = reinterpret_cast<float*>(s_x);                        // [FR][fs]   windowed frames (aliases s_x / s_z)
    __shared__ float2 s_tw[16][16];                                    // per-lane twiddles of the 16 x 16 FFT: [k2][lane]
    __shared__ float2 s_awin[MODE == 3 ? 16 : 1][16];                  // mode 3: analysis window [n2][lane] (samples 2 ln + 32 n2 ..)
    float* s_wav = reinterpret_cast<float*>(s_x + FR * ZSTRIDE);       // mode 3: the tile's samples, behind the shared storage

    const int tid = threadIdx.x, f = tid >> 4, ln = tid & 15;
    const int S = a.hop, L = a.frame_len, T = a.num_frames, F = a.num_bins;
    const int fs = (L + 3) & ~3;                                       // pitch of a windowed frame in LDS
    const float* __restrict__ tab = a.table;
    s_tw[tid >> 4][tid & 15] = *reinterpret_cast<const float2*>(tab + TAB_TW256 + 2 * (((tid & 15) * (tid >> 4)) & 255));
    if (MODE == 3) s_awin[tid >> 4][tid & 15] = *reinterpret_cast<const float2*>(tab + TAB_AWIN + 2 * (tid & 15) + 32 * (tid >> 4));
    const bool have_norm = a.mean != nullptr;

    // one spectrum element from the raw loaded values (r0 .. r3 as loaded by `fetch` below)
    auto make_x = [&](int kk, float r0, float r1, float r2, float r3, float mean_k, float std_k) -> cf {
        cf x{0.f, 0.f};
        if (MODE == 0) {          // complex spectrogram, interleaved
            x = {r0, r1};
        } else if (MODE == 1) {   // magnitude + phase planes
            float sn, cs;
            __sincosf(r1, &sn, &cs);
            x = {r0 * cs, r0 * sn};
        } else {                    // fused enhanced_sources: r0 = prediction, (r1, r2) = target STFT, r3 = mask
            float m = r0;
            if (have_norm) m = m * std_k + mean_k;
            m = __expf(m);
            if (HAS_MASK) {
                // models.py:186 casts the mask to complex64 and takes a FULL complex product
                // (a + bj)(m + 0j) = (a m - b 0) + (a 0 + b m) j; in a gap the signed zeros
                // decide tf.angle = atan2: (-0, +0) -> pi, everything else -> 0.  Kept as is.
                // (1 / |S| by v_rsq_f32: the kernel is bound by its instruction count -- ~4000 VALU instructions per wave and
                //  tile at two waves per SIMD -- and sqrt + two IEEE divisions per bin were a quarter of them.  The squares are
                //  scaled so that neither the 1e9-sized products of int16-range audio overflow nor tiny bins flush to zero.)
                const float zero = 0.f;
                const float pr = r1 * r3 - r2 * zero, pi = r1 * zero + r2 * r3;
                const float big = fmaxf(fabsf(pr), fabsf(pi));
                if (big > 0.f) {
                    const float sc = __builtin_amdgcn_rcpf(big), a = pr * sc, b2 = pi * sc;      // max(|a|, |b|) = 1
                    const float inv = __builtin_amdgcn_rsqf(a * a + b2 * b2);
                    x = {m * (a * inv), m * (b2 * inv)};
                } else {
                    x = {(__builtin_signbitf(pr) && !__builtin_signbitf(pi)) ? -m : m, 0.f};
                }
            } else {
                const float big = fmaxf(fabsf(r1), fabsf(r2));
                if (big > 0.f) {
                    const float sc = __builtin_amdgcn_rcpf(big), a = r1 * sc, b2 = r2 * sc;
                    const float inv = __builtin_amdgcn_rsqf(a * a + b2 * b2);
                    x = {m * (a * inv), m * (b2 * inv)};
                } else {
                    x = {__builtin_signbitf(r1) && !__builtin_signbitf(r2) ? -m : m, 0.f};  // atan2(+0,-0) = pi
                }
            }
        }
        if (kk == 0 || kk == 256) x.i = 0.f;  // irfft ignores the imaginary part of DC / Nyquist
        return x;
    };
    auto fetch = [&](int b, int t, int k, float& r0, float& r1, float& r2, float& r3) {
        const int64_t o = (int64_t)b * a.in_stride_b + (int64_t)t * a.in_stride_t;
        if (MODE == 0) {
            r0 = a.in0[o + 2 * k], r1 = a.in0[o + 2 * k + 1];
        } else if (MODE == 1) {
            r0 = a.in0[o + k], r1 = a.in1[o + k];
        } else {
            r0 = a.in0[o + k];
            if (MODE == 2) {
                const int64_t os = (int64_t)b * a.in1_stride_b + (int64_t)t * a.in1_stride_t;
                r1 = a.in1[os + 2 * k], r2 = a.in1[os + 2 * k + 1];
            }
            if (HAS_MASK) r3 = a.in2[(int64_t)b * a.in2_stride_b + (int64_t)t * a.in2_stride_t + k];
        }
    };
    __syncthreads();

    // (LDS-only barriers inside the tile loop: __syncthreads() also drains vmcnt, i.e. waits for the previous tile's output
    //  stores and, in mode 3, holds this tile's spectrum loads back behind them)
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_utt;
        const int h0 = (tile - b * tiles_per_utt) * (FR - 1);  // first output hop of the tile
        const int t0 = h0 - 1;                                  // first frame (halo)

        // ---- 1. spectrum tile -> LDS (thread <-> bin, coalesced rows), assembling X on the fly.  All 16 frames'
        //         loads are issued before anything is computed (they used to be consumed frame by frame, one
        //         memory latency per frame); `step` = 512 / fft length: a 256-point spectrum sits on the even
        //         bins of the 512 grid (the odd bins are zero), which makes the 512-point inverse 256-periodic.
        {
            const int kk = tid, k = kk / step;
            const bool kok = (kk % step == 0) && k < F;
            const float mean_k = (have_norm && kok) ? a.mean[k] : 0.f, std_k = (have_norm && kok) ? a.stdev[k] : 1.f;
            float r0[FR], r1[FR], r2[FR], r3[FR];
            if (MODE == 3) {
                // ---- 1a. the 16 frames' samples [t0 S, t0 S + 15 S + L) -> LDS, zero outside [0, wav_samples)
                const int seg = (FR - 1) * S + L;
                const int64_t s0 = (int64_t)t0 * S, NW = a.wav_samples;
                const float* src = a.wav + (int64_t)b * a.wav_stride_b;
                // whole 16-byte groups lie inside or outside the signal (wave-uniform test)
                const bool vec = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((S & 3) == 0) && ((NW & 3) == 0) && ((seg & 3) == 0);
                // (32 zeros behind the segment: the last column group of a frame whose length is no multiple of 32 reads them
                //  under zero window taps, and stale LDS could hold a NaN)
                if (vec) {
                    for (int i = 4 * tid; i < seg + 32; i += 4 * TPB) {
                        const int64_t n = s0 + i;
                        const bool ok = n >= 0 && n + 3 < NW && i < seg;
                        const float4 v4 = *reinterpret_cast<const float4*>(src + (ok ? n : 0));
                        *reinterpret_cast<float4*>(s_wav + i) = ok ? v4 : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                } else {
                    for (int i = tid; i < seg + 32; i += TPB) {
                        const int64_t n = s0 + i;
                        const bool ok = n >= 0 && n < NW && i < seg;
                        const float x = src[ok ? n : 0];
                        s_wav[i] = ok ? x : 0.f;
                    }
                }
            }
            float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 1.f;     // the Nyquist column: one frame per thread (tid < FR)
            const int tn = t0 + (tid & (FR - 1)), kn = 256 / step;
            const bool nok = tid < FR && (256 % step == 0) && kn < F && tn >= 0 && tn < T;
            auto fetch_tile = [&]() {
#pragma unroll
                for (int ff = 0; ff < FR; ++ff) {
                    // BRANCH-FREE: a load inside a per-lane `if` is closed by the compiler with s_waitcnt vmcnt(0) -- the 16
                    // frames were fetched one memory latency after the other.  Out-of-range lanes read element (0, 0)
                    // of the utterance instead; their value is never used (the select below).
                    const int t = t0 + ff;
                    const bool ok = kok && t >= 0 && t < T;
                    r0[ff] = r1[ff] = r2[ff] = 0.f, r3[ff] = 1.f;
                    fetch(b, ok ? t : 0, ok ? k : 0, r0[ff], r1[ff], r2[ff], r3[ff]);
                }
                if (tid < FR) fetch(b, nok ? tn : 0, nok ? kn : 0, q0, q1, q2, q3);
            };
            if (!(MODE == 3 && LATE)) fetch_tile();
            if (MODE == 3) {
                // ---- 1b. forward transform of the 16 frames (frontend.hip steps 2 - 4: window, 16 x 16 FFT of
                //          z[n] = x[2n] + j x[2n+1], natural-order Z with Z[256] := Z[0]); one frame per 16-lane group
                AVSI_LDS_BARRIER();
                cf v[16];
                {
                    const float* fr = s_wav + f * S + 2 * ln;
#pragma unroll
                    for (int n2 = 0; n2 < 16; ++n2) {
                        if (32 * n2 < L) {        // wave-uniform
                            const float2 x = *reinterpret_cast<const float2*>(fr + 32 * n2);
                            const float2 wn = s_awin[n2][ln];
                            v[n2] = {x.x * wn.x, x.y * wn.y};
                        } else {
                            v[n2] = {0.f, 0.f};
                        }
                    }
                }
                fft16(v);
                cf* zf = s_z + f * ZSTRIDE;
                zf[ln] = v[pos16(0)];
#pragma unroll
                for (int k2 = 1; k2 < 16; ++k2) {
                    const float2 w = s_tw[k2][ln];
                    zf[k2 * 17 + ln] = cmul(v[pos16(k2)], cf{w.x, w.y});
                }
                asm volatile("" ::: "memory");      // a frame lives in one wave: its LDS accesses execute in order
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) v[n1] = zf[ln * 17 + n1];
                asm volatile("" ::: "memory");
                fft16(v);
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) zf[16 * k1 + ln] = v[pos16(k1)];
                if (ln == 0) zf[256] = v[pos16(0)];
                AVSI_LDS_BARRIER();
                if (LATE) fetch_tile();
                // ---- 1c. thread <-> bin: S[k] = E[k] + W512^k O[k] from Z[k] and Z[256 - k] (frontend.hip step 5)
                const float2 wk = *reinterpret_cast<const float2*>(tab + TAB_TW512 + 2 * kk);
#pragma unroll
                for (int ff = 0; ff < FR; ++ff) {
                    const cf zk = s_z[ff * ZSTRIDE + kk], zm = s_z[ff * ZSTRIDE + 256 - kk];
                    const cf e{0.5f * (zk.r + zm.r), 0.5f * (zk.i - zm.i)};
                    const cf o{0.5f * (zk.i + zm.i), -0.5f * (zk.r - zm.r)};
                    r1[ff] = e.r + (o.r * wk.x - o.i * wk.y);
                    r2[ff] = e.i + (o.r * wk.y + o.i * wk.x);
                }
                if (tid < FR) {     // S[256] = Re Z[0] - Im Z[0], real
                    const cf z0 = s_z[tid * ZSTRIDE];
                    q1 = z0.r - z0.i, q2 = 0.f;
                }
                AVSI_LDS_BARRIER();    // every Z has been read: the storage becomes the spectrum tile
            }
#pragma unroll
            for (int ff = 0; ff < FR; ++ff) {
                const int t = t0 + ff;
                s_x[ff * XS + kk] = (kok && t >= 0 && t < T) ? make_x(kk, r0[ff], r1[ff], r2[ff], r3[ff], mean_k, std_k) : cf{0.f, 0.f};
            }
            if (tid < FR)
                s_x[tid * XS + 256] = nok ? make_x(256, q0, q1, q2, q3, have_norm ? a.mean[kn] : 0.f, have_norm ? a.stdev[kn] : 1.f)
                                          : cf{0.f, 0.f};
        }
        AVSI_LDS_BARRIER();

        // ---- 2. Z[k] = E[k] + j O[k] (conjugated for the forward-FFT trick), first 16-point FFT
        cf v[16];
        {
            const cf* xf = s_x + f * XS;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                const int k = ln + 16 * k2;
                const cf xk = xf[k], xm = xf[256 - k];
                const float2 w = *reinterpret_cast<const float2*>(tab + TAB_TW512 + 2 * k);  // e^{-2 pi j k/512}
                const cf e{0.5f * (xk.r + xm.r), 0.5f * (xk.i - xm.i)};
                const cf d{0.5f * (xk.r - xm.r), 0.5f * (xk.i + xm.i)};
                const cf o{d.r * w.x + d.i * w.y, d.i * w.x - d.r * w.y};  // d * conj(w) = d e^{+2 pi j k/512}
                // Z = E + j O ; feed conj(Z)
                v[k2] = {e.r - o.i, -(e.i + o.r)};
            }
        }
        fft16(v);
        AVSI_LDS_BARRIER();        // every lane has read its s_x inputs: the storage becomes s_z
        cf* zf = s_z + f * ZSTRIDE;
        zf[ln] = v[pos16(0)];
#pragma unroll
        for (int n2 = 1; n2 < 16; ++n2) {
            const float2 w = s_tw[n2][ln];
            zf[n2 * 17 + ln] = cmul(v[pos16(n2)], cf{w.x, w.y});
        }
        AVSI_LDS_BARRIER();
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) v[k1] = zf[ln * 17 + k1];
        AVSI_LDS_BARRIER();
        fft16(v);
        // ---- 3. z[16 n1 + ln] = conj(.) / 256 -> x[2n], x[2n+1]; synthesis window; store frame
        {
            float* ffr = s_f + f * fs;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int n = 16 * n1 + ln;
                const cf z = v[pos16(n1)];
                const float2 w = *reinterpret_cast<const float2*>(tab + TAB_WIN + 2 * n);
                const float sc = (float)step * (1.f / 256.f);
                if (2 * n < L) *reinterpret_cast<float2*>(ffr + 2 * n) = make_float2(z.r * sc * w.x, -z.i * sc * w.y);
            }
        }
        AVSI_LDS_BARRIER();

        // ---- 4. overlap-add: every sample of the 15 hops sums the frames that cover it, stored once
        // (frame_len <= 2 hop: a sample is covered by the frame of its own hop and the one before -- the halo frame for the
        //  tile's first hop.  Hop by hop, lanes along the hop: no integer division per sample.)
        const int64_t n_out = a.num_samples;
        float* orow = a.out + (int64_t)b * a.out_stride_b;
#pragma unroll 1
        for (int hop_i = 0; hop_i < FR - 1; ++hop_i) {
            const int h = h0 + hop_i;
            if (h >= n_hops) break;
            const float* f1 = s_f + (hop_i + 1) * fs;       // frame t = h
            const float* f0 = s_f + hop_i * fs + S;         // frame t = h - 1, second part
            for (int r = tid; r < S; r += TPB) {
                const int64_t n = (int64_t)h * S + r;
                if (n >= n_out) break;
                float acc = f1[r];
                if (r + S < L) acc += f0[r];
                orow[n] = acc;
            }
        }
        AVSI_LDS_BARRIER();
    }
}

}  // namespace

extern "C" size_t avsi_istft_table_floats(int frame_len, int hop, int nfft) {
    if ((nfft != 512 && nfft != 256) || frame_len <= 0 || frame_len > nfft || hop <= 0) return 0;
    return TAB_FLOATS;
}

extern "C" int avsi_istft_init_tables(float* table, int frame_len, int hop, int nfft, void* stream) {
    if (!table) return AVSI_ERR_INVALID_ARG;
    if ((nfft != 512 && nfft != 256) || frame_len <= 0 || frame_len > nfft || (frame_len & 1) || hop <= 0 || hop > frame_len)
        return AVSI_ERR_UNSUPPORTED;
    avsi_clear_error();
    hipLaunchKernelGGL(istft_tables_kernel, dim3(3), dim3(256), 0, (hipStream_t)stream, table, frame_len, hop);
    return avsi_launch_status();
}

extern "C" int avsi_istft_f32(const avsi_istft_args* args, void* stream) {
    if (!args) return AVSI_ERR_INVALID_ARG;
    const avsi_istft_args& a = *args;
    if (!a.in0 || !a.out || !a.table || a.batch <= 0 || a.num_frames <= 0 || a.num_samples <= 0)
        return AVSI_ERR_INVALID_ARG;
    if (a.mode < 0 || a.mode > 3 || ((a.mode == 1 || a.mode == 2) && !a.in1)) return AVSI_ERR_INVALID_ARG;
    if (a.mode == 3 && (!a.wav || a.wav_samples <= 0)) return AVSI_ERR_INVALID_ARG;
    if ((a.mean == nullptr) != (a.stdev == nullptr)) return AVSI_ERR_INVALID_ARG;
    if ((a.nfft != 512 && a.nfft != 256) || a.frame_len <= 0 || a.frame_len > a.nfft || (a.frame_len & 1) ||
        a.hop <= 0 || a.hop > a.frame_len || a.frame_len > 2 * a.hop)
        return AVSI_ERR_UNSUPPORTED;  // the 16-frame tile carries ONE halo frame: frame_len <= 2 hop
    if (a.num_bins <= 0 || a.num_bins > a.nfft / 2 + 1) return AVSI_ERR_INVALID_ARG;
    const int step = 512 / a.nfft;
    const int64_t full = (int64_t)(a.num_frames - 1) * a.hop + a.frame_len;
    if (a.num_samples > full) return AVSI_ERR_INVALID_ARG;
    const int n_hops = (int)avsi_ceil_div(a.num_samples, a.hop);
    const int tiles_per_utt = (int)avsi_ceil_div(n_hops, FR - 1);
    const int64_t n_tiles64 = (int64_t)a.batch * tiles_per_utt;
    if (n_tiles64 > INT32_MAX) return AVSI_ERR_UNSUPPORTED;
    // spectrum tile, FFT scratch and windowed frames share one region (ZSTRIDE >= XS; 16 frames of <= 512 floats fit in it);
    // mode 3 keeps the tile's samples behind it
    size_t lds = (size_t)FR * ZSTRIDE * 8;
    if (lds < (size_t)FR * ((a.frame_len + 3) & ~3) * 4) return AVSI_ERR_UNSUPPORTED;
    if (a.mode == 3) lds += (size_t)((FR - 1) * a.hop + a.frame_len + 32) * 4;
    const int n_tiles = (int)n_tiles64;
    const int per_cu = 3;                                    // workgroups per CU: 35 - 48 KB of LDS, <= 168 registers
    const int grid = n_tiles < AVSI_NUM_CU * per_cu ? n_tiles : AVSI_NUM_CU * per_cu;
    avsi_clear_error();
#define AVSI_ISTFT_LAUNCH(...)                                                                                          \
    do {                                                                                                                \
        (void)hipFuncSetAttribute((const void*)istft_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((istft_kernel<__VA_ARGS__>), dim3(grid), dim3(TPB), lds, (hipStream_t)stream, a, tiles_per_utt, n_tiles, \
                           n_hops, step);                                                                               \
    } while (0)
    if (a.mode == 0) AVSI_ISTFT_LAUNCH(0, false);
    else if (a.mode == 1) AVSI_ISTFT_LAUNCH(1, false);
    else if (a.mode == 2 && a.in2) AVSI_ISTFT_LAUNCH(2, true);
    else if (a.mode == 2) AVSI_ISTFT_LAUNCH(2, false);
    else if (a.in2) AVSI_ISTFT_LAUNCH(3, true);
    else AVSI_ISTFT_LAUNCH(3, false);
#undef AVSI_ISTFT_LAUNCH
    return avsi_launch_status();
}
