// Fused speech front end for gfx950 (MI355X): framing + periodic-Hann window + 512-point real
// FFT + |X|^p / log / z-norm / mask (+ sparse-triangular log-mel), one pass over HBM.
//
// Replaces tf.contrib.signal.stft (reference audio_processing.py:35-36), get_spectrogram
// (:45-56), get_log_mel_spectrogram (:59-72) and the elementwise chain of
// StackedBLSTMModel.__init__ (models.py:31-35).
//
// Work decomposition (wave = 64 lanes):
//   - a tile = 16 consecutive frames of one utterance; a 256-thread workgroup owns a tile and
//     walks tiles persistently (tile = blockIdx.x, += gridDim.x);
//   - the tile's samples ((16-1)*hop + frame_len floats) are read from HBM once with 16-byte
//     loads into LDS, so the 50 % frame overlap is served from LDS, not re-read;
//   - the 512-point real FFT of a frame is a 256-point complex FFT of z[n] = x[2n] + j x[2n+1],
//     factored 16 x 16: each 16-lane group owns one frame, each lane runs a 16-point FFT
//     entirely in registers, one transposition through LDS, a second in-register 16-point FFT;
//   - the epilogue maps thread -> frequency bin (so per-bin twiddle / mean / std stay in
//     registers and every global store is a full coalesced row), recovers X[k] from Z[k] and
//     Z[256-k], and writes every requested output exactly once.
#include <stdlib.h>

#include "avsi_common.h"
#include "fft16.h"

namespace {

constexpr int FR = 16;        // frames per tile
constexpr int TPB = 256;      // threads per workgroup
constexpr int ZSTRIDE = 272;  // complex elements per frame in LDS: 16 rows x 17 (bank padding)
constexpr int PSTRIDE = 260;  // floats per frame of the power-spectrum tile

using namespace avsi_fft;

// table layout (floats): window[512] (zero past frame_len) | W256^m (m<256) as (re,im) | W512^k (k<=256)
constexpr int TAB_WIN = 0;
constexpr int TAB_TW256 = 512;
constexpr int TAB_TW512 = 512 + 512;
constexpr int TAB_FLOATS = 512 + 512 + 2 * 257 + 2;

__global__ void frontend_tables_kernel(float* tab, int frame_len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 512) {
        // periodic Hann: 0.5 - 0.5 cos(2 pi n / L)
        tab[TAB_WIN + i] = i < frame_len ? (float)(0.5 - 0.5 * cospi(2.0 * i / frame_len)) : 0.f;
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

// NB = number of 32-sample column groups that can hold non-zero window taps = ceil(frame_len/32)
// WR = float4 staging registers per thread for the next tile's samples (256 threads x 4 floats each: WR = 4 covers
// the 15 x 192 + 384 samples of the standard geometry, 8 covers hop <= 512); OCC = workgroups per CU the register
// allocation is held to (3 -> 168 VGPRs).
// STEP1: nfft = 512 (column = bin: the column / liveness arithmetic of the 256-point case folds away).
// MODEL: the inpainter's own call (models.py:30-35) -- nfft 512, >= 256 bins, normalised log-magnitude spectrum AND
// masked features out, mask given, no complex / log-mel output: every store and every mask load of the epilogue is
// unconditional, so the compiler sees straight-line code and can count the stores it may leave in flight.
// MD (round 5): 0 = the generic kernel; 1 = MODEL as above; 2 = MODEL without the un-masked target output -- the masked
// features are the ONLY store, the kernel moves exactly its 706 kB per utterance; 3 = the L1 loss of the step taken HERE
// (avsi_frontend_l1_loss_f32): the same transform of the same samples, but instead of storing anything the epilogue loads
// the prediction where the target would have gone and accumulates sum|t - p|, sum|t - p|(1 - m), sum(1 - m), sum|t - p| m,
// sum m (and stores d loss / d prediction when asked).  Together, modes 2 and 3 replace "write the target (257 kB per
// utterance) at the start of the step, read it back in the loss kernel at its end" by a second read of the waveform (192 kB).
struct FeLoss {
    const float* pred;        // [B][T][257] prediction (mode 3)
    int64_t stride_b, stride_t;
    float* dpred;             // same layout, optional: sign(p - t) * gscale
    float gscale;
    float* part;              // [gridDim.x][5] partial sums
};

template <int NB, int WR, int OCC, bool STEP1, int MD>
__global__ __launch_bounds__(TPB, OCC) void frontend_kernel(const avsi_frontend_args a, const int tiles_per_utt,
                                                         const int n_tiles, const int seg_floats, const int step_arg,
                                                         const FeLoss lo) {
    constexpr bool MODEL = MD != 0;
    constexpr bool LOSS = MD == 3;
    const int step = STEP1 ? 1 : step_arg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_wav = reinterpret_cast<float*>(smem);
    cf* s_z = reinterpret_cast<cf*>(smem + (size_t)(seg_floats + 4) * 4);
    float* s_pow = reinterpret_cast<float*>(smem + (size_t)(seg_floats + 4) * 4 + (size_t)FR * ZSTRIDE * 8);

    const int tid = threadIdx.x;
    const int f = tid >> 4;   // frame of the tile handled in the FFT phase
    const int ln = tid & 15;  // lane within the frame's 16-lane group
    const int S = a.hop, N = a.num_samples, T = a.num_frames, F = a.num_bins;
    const float* __restrict__ tab = a.table;

    // ---- per-lane constants of the FFT phase live in LDS ([index][lane], conflict-free float2 reads):
    //      as registers they were 54 VGPRs of a kernel that needed 317 and therefore ran one wave per SIMD
    __shared__ float2 s_win[16][16], s_tw[16][16];
    {
        const int i = tid >> 4, l = tid & 15;     // 256 threads = 16 x 16 entries each
        s_win[i][l] = i < NB ? *reinterpret_cast<const float2*>(tab + TAB_WIN + 2 * l + 32 * i) : make_float2(0.f, 0.f);
        s_tw[i][l] = *reinterpret_cast<const float2*>(tab + TAB_TW256 + 2 * ((l * i) & 255));
    }
    __syncthreads();
    // ---- epilogue mapping: thread <-> bins kq, kq + 64, kq + 128, kq + 192 x 4 frames (its wave's quarter of the
    //      tile): a wave reads 64 CONSECUTIVE complex values of the LDS row per access (and 64 consecutive ones of
    //      the mirrored half, Z[256 - k]) -- no bank conflicts -- and every global access is 256 contiguous bytes
    //      per wave.  (Four consecutive bins per thread gave 16-byte global accesses but a 32-byte lane stride in
    //      LDS: 4-way conflicts on every read of the spectrum tile.)  The per-bin constants (twiddle, mean, 1/std)
    //      sit in LDS as one float4 per thread each and are fetched at the top of the epilogue: as registers they
    //      were 16 VGPRs alive across the whole FFT phase of a kernel whose occupancy is set by its registers.
    const int kq = tid & 63, fg = tid >> 6;
    const bool have_norm = a.mean != nullptr;
    __shared__ float4 s_c[4][64];          // [twiddle re | twiddle im | mean | 1 / std][kq]
    // `step` = 512 / nfft: a 256-point transform is read off the even bins of the 512-point one (the
    // frame is zero-padded to 512, which interpolates the spectrum), so output column = bin / step.
    bool live[4];
    int col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        col[j] = STEP1 ? kq + 64 * j : (kq + 64 * j) / step;
        live[j] = STEP1 ? (kq + 64 * j < F) : (((kq + 64 * j) % step == 0) && col[j] < F);
    }
    {
        float wr4[4], wi4[4], mean4[4], istd4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 w = *reinterpret_cast<const float2*>(tab + TAB_TW512 + 2 * (kq + 64 * j));
            wr4[j] = w.x, wi4[j] = w.y;
            mean4[j] = (have_norm && live[j]) ? a.mean[col[j]] : 0.f;
            istd4[j] = (have_norm && live[j]) ? 1.f / a.stdev[col[j]] : 1.f;
        }
        if (tid < 64) {
            s_c[0][kq] = make_float4(wr4[0], wr4[1], wr4[2], wr4[3]);
            s_c[1][kq] = make_float4(wi4[0], wi4[1], wi4[2], wi4[3]);
            s_c[2][kq] = make_float4(mean4[0], mean4[1], mean4[2], mean4[3]);
            s_c[3][kq] = make_float4(istd4[0], istd4[1], istd4[2], istd4[3]);
        }
    }
    __syncthreads();
    const int col_n = 256 / step;                                        // Nyquist bin
    const float mean_n = (have_norm && F > col_n) ? a.mean[col_n] : 0.f;
    const float istd_n = (have_norm && F > col_n) ? 1.f / a.stdev[col_n] : 1.f;
    const bool want_pow = !MODEL && a.out_logmel != nullptr;
    const bool has_stft = !MODEL && a.out_stft != nullptr;
    const bool has_spec = MD == 1 || (!MODEL && a.out_spec != nullptr);
    const bool has_feat = MODEL || a.out_feat != nullptr;        // (mode 3 stores no features but wants their mask values)
    const bool has_mask = MODEL || a.mask != nullptr;
    const float spec_power = MODEL ? 1.f : a.spec_power;
    const bool log_spec = MODEL || a.log_spec;

    // Software pipeline over tiles: the NEXT tile's samples are fetched into registers while the
    // current tile is transformed, and the current tile's mask values are fetched before the FFT and
    // consumed after it, so neither HBM latency sits on the critical path of a tile.
    float ls_all = 0.f, ls_hole = 0.f, ln_hole = 0.f, ls_valid = 0.f, ln_valid = 0.f;      // mode 3: this thread's partial sums
    float4 wreg[WR];
    // BRANCH-FREE on purpose: with a per-lane `if (in range) load` the loads sat in divergent blocks and the
    // compiler closed every one of them with s_waitcnt vmcnt(0) -- four serialised HBM round trips per tile
    // (~8 of the 12 us a tile took), and nothing of the intended prefetch.  Out-of-range lanes load the tile's
    // first sample(s) instead and select zero.
    auto fetch_wav = [&](int tl) {
        const int bb = tl / tiles_per_utt;
        const int64_t s0 = (int64_t)(tl - bb * tiles_per_utt) * FR * S;
        const float* src = a.wav + (int64_t)bb * a.wav_stride + s0;
        const int valid = (int)max((int64_t)0, min((int64_t)seg_floats, (int64_t)N - s0));      // >= 1
        // wave-uniform: whole float4s only (aligned rows, and the samples of this tile end on a multiple of four)
        const bool vec = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((valid & 3) == 0);
        if (vec) {
#pragma unroll
            for (int p = 0; p < WR; ++p) {
                const int i = (p * TPB + tid) * 4;
                const bool ok = i + 3 < valid;
                const float4 v = *reinterpret_cast<const float4*>(src + (ok ? i : 0));
                wreg[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int p = 0; p < WR; ++p) {
                const int i = (p * TPB + tid) * 4;
                const float x0 = src[i < valid ? i : 0], x1 = src[i + 1 < valid ? i + 1 : 0];
                const float x2 = src[i + 2 < valid ? i + 2 : 0], x3 = src[i + 3 < valid ? i + 3 : 0];
                wreg[p] = make_float4(i < valid ? x0 : 0.f, i + 1 < valid ? x1 : 0.f, i + 2 < valid ? x2 : 0.f,
                                      i + 3 < valid ? x3 : 0.f);
            }
        }
    };
    if (blockIdx.x < n_tiles) fetch_wav(blockIdx.x);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_utt;
        const int t0 = (tile - b * tiles_per_utt) * FR;

        // ---- 1. staged samples: registers -> LDS (zero past the signal end), then refill the registers
#pragma unroll
        for (int p = 0; p < WR; ++p) {
            const int i = (p * TPB + tid) * 4;
            // lanes past the segment write a spare 16-byte slot behind it (no per-lane branch around the access)
            *reinterpret_cast<float4*>(s_wav + (i < seg_floats ? i : seg_floats)) = wreg[p];
        }
        // LDS-only barriers throughout the tile loop: __syncthreads() would also drain vmcnt, i.e. wait for this
        // tile's output stores and for the NEXT tile's sample / mask loads -- exactly what the pipeline keeps in flight
        AVSI_LDS_BARRIER();
        if (tile + (int)gridDim.x < n_tiles) fetch_wav(tile + gridDim.x);
        // ---- 2. window + first 16-point FFT (over n2, with n = ln + 16 n2) + twiddle
        cf v[16];
        {
            const float* fr = s_wav + f * S + 2 * ln;
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) {
                if (n2 < NB) {
                    const float2 x = *reinterpret_cast<const float2*>(fr + 32 * n2);
                    const float2 wn = s_win[n2][ln];
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
        // ---- 3. transposition: lane k2 = ln gathers its 16 n1 values.  A frame lives in ONE wave
        //         (16 consecutive lanes) and a wave's LDS accesses execute in order, so no
        //         workgroup barrier is needed here -- only the compiler must not reorder.
        asm volatile("" ::: "memory");
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = zf[ln * 17 + n1];
        asm volatile("" ::: "memory");
        // this tile's mask values: requested here, in flight during the second FFT and the barrier, consumed by the
        // epilogue (requested before the first FFT they were 16 more registers alive across both)
        float mk[4][4], mkn[4];
        float pp[4][4], ppn[4];                    // mode 3: the prediction at this thread's bins, fetched like the mask
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            const int t = t0 + fg * 4 + fi;
            mkn[fi] = 1.f;
            ppn[fi] = 0.f;
            if (MODEL && t < T)       // Nyquist column of the mask (one address per frame: a broadcast load)
                mkn[fi] = a.mask[(int64_t)b * a.mask_stride_b + (int64_t)t * a.mask_stride_t + 256];
            if (LOSS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) pp[fi][j] = 0.f;
                if (t < T) {
                    const float* pr = lo.pred + (int64_t)b * lo.stride_b + (int64_t)t * lo.stride_t;
                    ppn[fi] = pr[256];
#pragma unroll
                    for (int j = 0; j < 4; ++j) pp[fi][j] = pr[kq + 64 * j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) mk[fi][j] = 1.f;
            if (has_feat && has_mask && t < T) {       // wave-uniform; the loads themselves branch-free
                const float* mp = a.mask + (int64_t)b * a.mask_stride_b + (int64_t)t * a.mask_stride_t;
#pragma unroll
                for (int j = 0; j < 4; ++j) mk[fi][j] = mp[MODEL ? kq + 64 * j : (live[j] ? col[j] : 0)];      // RAW: any arithmetic on the value here
                                                                               // (even a select) makes the wave wait for it now
            }
        }


        // ---- 4. second 16-point FFT (over n1): Z[16 k1 + ln], natural order; Z[256] := Z[0]
        fft16(v);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) zf[16 * k1 + ln] = v[pos16(k1)];
        if (ln == 0) zf[256] = v[pos16(0)];
        AVSI_LDS_BARRIER();

        // ---- 5. epilogue: X[k] = E[k] + W512^k O[k],  E = (Z[k] + conj Z[256-k]) / 2,
        //         O = -j (Z[k] - conj Z[256-k]) / 2
        // The mask values are needed from here on.  Pin the wait for them HERE, before any store of this tile is
        // issued: vmcnt counts loads and stores in order, and once stores sit between the mask loads and their
        // first use the only wait the compiler can prove is vmcnt(0) -- a full drain of those stores, per frame.
        if (has_feat) {
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mov_b32 %0, %0" : "+v"(mk[fi][j])::"memory");
            if (MODEL) {
#pragma unroll
                for (int fi = 0; fi < 4; ++fi) asm volatile("v_mov_b32 %0, %0" : "+v"(mkn[fi])::"memory");
            }
            if (LOSS) {
#pragma unroll
                for (int fi = 0; fi < 4; ++fi) {
                    asm volatile("v_mov_b32 %0, %0" : "+v"(ppn[fi])::"memory");
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("v_mov_b32 %0, %0" : "+v"(pp[fi][j])::"memory");
                }
            }
        }
        const bool all_live = MODEL || (STEP1 && F >= 256);        // wave-uniform: every bin of every lane is stored
        const float4 wkr4 = s_c[0][kq], wki4 = s_c[1][kq], mean4v = s_c[2][kq], istd4v = s_c[3][kq];
        const float wkr[4] = {wkr4.x, wkr4.y, wkr4.z, wkr4.w}, wki[4] = {wki4.x, wki4.y, wki4.z, wki4.w};
        const float mean_k[4] = {mean4v.x, mean4v.y, mean4v.z, mean4v.w}, istd_k[4] = {istd4v.x, istd4v.y, istd4v.z, istd4v.w};
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            const int ff = fg * 4 + fi;
            const int t = t0 + ff;
            if (t >= T) continue;
            const cf* zrow = s_z + ff * ZSTRIDE;
            float xr[4], xi[4], p2[4], sp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kq + 64 * j;
                const cf zk = zrow[k], zm = zrow[256 - k];
                const cf e{0.5f * (zk.r + zm.r), 0.5f * (zk.i - zm.i)};
                const cf o{0.5f * (zk.i + zm.i), -0.5f * (zk.r - zm.r)};
                xr[j] = e.r + (o.r * wkr[j] - o.i * wki[j]);
                xi[j] = e.i + (o.r * wki[j] + o.i * wkr[j]);
                p2[j] = xr[j] * xr[j] + xi[j] * xi[j];
                float sv = __builtin_amdgcn_sqrtf(p2[j]);
                if (spec_power != 1.f) sv = (spec_power == 2.f) ? sv * sv : __powf(sv, spec_power);
                if (log_spec) sv = __logf(sv + a.eps);
                sp[j] = (sv - mean_k[j]) * istd_k[j];
            }
            if (want_pow) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s_pow[ff * PSTRIDE + kq + 64 * j] = p2[j];
            }
            float* o2 = a.out_stft + (int64_t)b * a.stft_stride_b + (int64_t)t * a.stft_stride_t;
            float* o1 = a.out_spec + (int64_t)b * a.spec_stride_b + (int64_t)t * a.spec_stride_t;
            float* o3 = a.out_feat + (int64_t)b * a.feat_stride_b + (int64_t)t * a.feat_stride_t;
            if (all_live) {     // straight-line stores (no per-lane branch: the compiler can count them)
                if (has_stft) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<float2*>(o2 + 2 * (kq + 64 * j)) = make_float2(xr[j], xi[j]);
                }
                if (has_spec) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o1[kq + 64 * j] = sp[j];
                }
                if (has_feat && !LOSS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o3[kq + 64 * j] = sp[j] * mk[fi][j];
                }
                if (MODEL) {
                    // Nyquist bin (X[256] = Re Z[0] - Im Z[0], real) and the zero fill of the padded feature columns,
                    // by the frame's own wave: lane 0 owns column 256, lanes 1 .. feat_cols - 257 the padding
                    const cf z0 = zrow[0];
                    const float svn = (__logf(fabsf(z0.r - z0.i) + a.eps) - mean_n) * istd_n;
                    if (MD == 1 && kq == 0) o1[256] = svn;
                    if (!LOSS && kq < a.feat_cols - 256) o3[256 + kq] = kq == 0 ? svn * mkn[fi] : 0.f;
                    if (LOSS) {
                        float* dp = lo.dpred ? lo.dpred + (int64_t)b * lo.stride_b + (int64_t)t * lo.stride_t : nullptr;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float d = pp[fi][j] - sp[j], e = fabsf(d), m = mk[fi][j];
                            ls_all += e, ls_hole += e * (1.f - m), ln_hole += 1.f - m, ls_valid += e * m, ln_valid += m;
                            if (dp) dp[kq + 64 * j] = d > 0.f ? lo.gscale : (d < 0.f ? -lo.gscale : 0.f);
                        }
                        if (kq == 0) {
                            const float d = ppn[fi] - svn, e = fabsf(d), m = mkn[fi];
                            ls_all += e, ls_hole += e * (1.f - m), ln_hole += 1.f - m, ls_valid += e * m, ln_valid += m;
                            if (dp) dp[256] = d > 0.f ? lo.gscale : (d < 0.f ? -lo.gscale : 0.f);
                        }
                    }
                }
            } else {
                if (a.out_stft) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (live[j]) *reinterpret_cast<float2*>(o2 + 2 * col[j]) = make_float2(xr[j], xi[j]);
                }
                if (a.out_spec) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (live[j]) o1[col[j]] = sp[j];
                }
                if (a.out_feat) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (live[j]) o3[col[j]] = sp[j] * mk[fi][j];
                }
            }
        }
        // Nyquist bin (k = 256) and zero fill of the padded feature columns: thread <-> frame
        if (!MODEL && tid < FR && t0 + tid < T) {
            const int t = t0 + tid;
            const cf z0 = s_z[tid * ZSTRIDE];
            const float xn = z0.r - z0.i;  // X[256] = Re Z[0] - Im Z[0], purely real
            if (want_pow) s_pow[tid * PSTRIDE + 256] = xn * xn;
            if (F > col_n) {
                if (a.out_stft) {
                    float* o2 = a.out_stft + (int64_t)b * a.stft_stride_b + (int64_t)t * a.stft_stride_t + 2 * col_n;
                    o2[0] = xn;
                    o2[1] = 0.f;
                }
                if (a.out_spec || a.out_feat) {
                    float sv = fabsf(xn);
                    if (a.spec_power != 1.f) sv = (a.spec_power == 2.f) ? sv * sv : __powf(sv, a.spec_power);
                    if (a.log_spec) sv = __logf(sv + a.eps);
                    sv = (sv - mean_n) * istd_n;
                    if (a.out_spec) a.out_spec[(int64_t)b * a.spec_stride_b + (int64_t)t * a.spec_stride_t + col_n] = sv;
                    if (a.out_feat) {
                        const float m =
                            a.mask ? a.mask[(int64_t)b * a.mask_stride_b + (int64_t)t * a.mask_stride_t + col_n] : 1.f;
                        a.out_feat[(int64_t)b * a.feat_stride_b + (int64_t)t * a.feat_stride_t + col_n] = sv * m;
                    }
                }
            }
            if (a.out_feat)
                for (int c = F; c < a.feat_cols; ++c)
                    a.out_feat[(int64_t)b * a.feat_stride_b + (int64_t)t * a.feat_stride_t + c] = 0.f;
        }

        // ---- 6. log-mel: sparse triangular bands over the power spectrum tile
        if (want_pow) {
            AVSI_LDS_BARRIER();
            const int items = FR * a.num_mel;
            for (int it = tid; it < items; it += TPB) {
                const int ff = it / a.num_mel, m = it - ff * a.num_mel;
                const int t = t0 + ff;
                if (t >= T) continue;
                const int st = a.mel_start[m], len = a.mel_len[m];
                const float* w = a.mel_w + (int64_t)m * a.mel_w_stride;
                const float* p = s_pow + ff * PSTRIDE + st;
                float acc = 0.f;
                for (int j = 0; j < len; ++j) acc = fmaf(p[j], w[j], acc);
                a.out_logmel[(int64_t)b * a.logmel_stride_b + (int64_t)t * a.logmel_stride_t + m] = __logf(acc + a.eps);
            }
        }
        AVSI_LDS_BARRIER();  // LDS is re-staged by the next tile
    }
    if (LOSS) {
        // one partial row per workgroup: wave shuffle trees, then the four waves in order (fixed order: deterministic)
        __shared__ float red[TPB / 64][5];
        float v5[5] = {ls_all, ls_hole, ln_hole, ls_valid, ln_valid};
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v5[k] += __shfl_xor(v5[k], o, 64);
        if ((tid & 63) == 0)
#pragma unroll
            for (int k = 0; k < 5; ++k) red[tid >> 6][k] = v5[k];
        __syncthreads();
        if (tid < 5) lo.part[blockIdx.x * 5 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

}  // namespace

extern "C" size_t avsi_frontend_table_floats(int frame_len, int nfft) {
    if ((nfft != 512 && nfft != 256) || frame_len <= 0 || frame_len > nfft) return 0;
    return TAB_FLOATS;
}

extern "C" int avsi_frontend_init_tables(float* table, int frame_len, int nfft, void* stream) {
    if (!table) return AVSI_ERR_INVALID_ARG;
    if ((nfft != 512 && nfft != 256) || frame_len <= 0 || frame_len > nfft || (frame_len & 1)) return AVSI_ERR_UNSUPPORTED;
    avsi_clear_error();
    hipLaunchKernelGGL(frontend_tables_kernel, dim3(3), dim3(256), 0, (hipStream_t)stream, table, frame_len);
    return avsi_launch_status();
}

// internal (loss.hip): out3 <- the five sums of `nblocks` partial rows (double, fixed order)
int avsi_l1_final_launch(const float* part, int nblocks, int64_t n, float* out3, hipStream_t st);

// mode: 0 = what the arguments ask for (avsi_frontend_f32); 3 = the loss form (avsi_frontend_l1_loss_f32)
static int frontend_launch(const avsi_frontend_args& a, int mode, const FeLoss& lo, int* grid_out, void* stream) {
    if (!a.wav || !a.table || a.batch <= 0 || a.num_samples <= 0 || a.wav_stride < a.num_samples)
        return AVSI_ERR_INVALID_ARG;
    if ((a.nfft != 512 && a.nfft != 256) || a.frame_len <= 0 || a.frame_len > a.nfft || (a.frame_len & 1) ||
        a.hop <= 0 || (a.hop & 1))
        return AVSI_ERR_UNSUPPORTED;
    if (a.nfft != 512 && a.out_logmel) return AVSI_ERR_UNSUPPORTED;  // mel bands are defined on the 512 grid
    const int step = 512 / a.nfft;
    const int t_full = (int)avsi_ceil_div(a.num_samples, a.hop);
    if (a.num_frames <= 0 || a.num_frames > t_full) return AVSI_ERR_INVALID_ARG;
    if (a.num_bins <= 0 || a.num_bins > a.nfft / 2 + 1) return AVSI_ERR_INVALID_ARG;
    if ((a.mean == nullptr) != (a.stdev == nullptr)) return AVSI_ERR_INVALID_ARG;
    if (a.out_feat && a.feat_cols < a.num_bins) return AVSI_ERR_INVALID_ARG;
    // complex bins are stored as 8-byte pairs
    if (a.out_stft && (((a.stft_stride_b | a.stft_stride_t) & 1) || (reinterpret_cast<uintptr_t>(a.out_stft) & 7)))
        return AVSI_ERR_INVALID_ARG;
    if (a.out_logmel) {
        if (a.num_mel <= 0 || !a.mel_start || !a.mel_len || !a.mel_w || a.mel_w_stride <= 0) return AVSI_ERR_INVALID_ARG;
    }
    if (mode != 3 && !a.out_stft && !a.out_spec && !a.out_feat && !a.out_logmel) return AVSI_OK;

    const int nb_need = (a.frame_len + 31) / 32;
    const int nb = nb_need <= 8 ? 8 : (nb_need <= 12 ? 12 : 16);  // template instance actually launched
    const int seg = (int)avsi_round_up((int64_t)(FR - 1) * a.hop + 32 * nb, 4);
    const size_t lds = (size_t)(seg + 4) * 4 + (size_t)FR * ZSTRIDE * 8 + (a.out_logmel ? (size_t)FR * PSTRIDE * 4 : 0);
    if (lds > 160 * 1024 || seg > 8 * TPB * 4) return AVSI_ERR_UNSUPPORTED;  // staging registers cover 8192 samples
    const int tiles_per_utt = (int)avsi_ceil_div(a.num_frames, FR);
    const int64_t n_tiles64 = (int64_t)a.batch * tiles_per_utt;
    if (n_tiles64 > INT32_MAX) return AVSI_ERR_UNSUPPORTED;
    const int n_tiles = (int)n_tiles64;
    // Resident workgroups per CU: TWO.  The model specialisation needs 165 VGPRs and would fit three, but three are
    // slower (1.97 vs 1.87 ms at 8192 utterances): with the serialised waits gone the kernel moves 4.3 TB/s of real
    // traffic (two outputs + mask + samples), and a third workgroup per CU only adds concurrent streams.
    // AVSI_FE_OCC=3: diagnostics.
    const bool small = seg <= 4 * TPB * 4;
    // the inpainter's own shapes (see the MD template flag); which outputs are asked for picks the mode
    const bool model_shape = small && step == 1 && a.num_bins == 257 && a.mask && a.mean && !a.out_stft && !a.out_logmel &&
                             a.spec_power == 1.f && a.log_spec;
    const bool feat_ok = a.out_feat && a.feat_cols >= 257 && a.feat_cols <= 256 + 64;
    const bool model = mode != 3 && model_shape && a.out_spec && feat_ok;
    const bool model_feat_only = mode != 3 && model_shape && !a.out_spec && feat_ok && nb == 12;     // mode 2: 24 ms frames only
    if (mode == 3 && !(model_shape && nb == 12)) return AVSI_ERR_UNSUPPORTED;
    static const bool env_occ3 = getenv("AVSI_FE_OCC") && atoi(getenv("AVSI_FE_OCC")) == 3;      // diagnostics, read once
    const int occ = (model && nb == 12 && env_occ3) ? 3 : 2;
    const int by_lds = (int)(156 * 1024 / (lds + 4352));       // + the static constant tables
    const int wg_per_cu = by_lds > occ ? occ : (by_lds < 1 ? 1 : by_lds);
    const int grid = n_tiles < AVSI_NUM_CU * wg_per_cu ? n_tiles : AVSI_NUM_CU * wg_per_cu;
    if (grid_out) *grid_out = grid;
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();

#define AVSI_FE_LAUNCH1(NBV, WRV, OCCV, S1V, MDV)                                                                               \
    do {                                                                                                               \
        if (lds > 64 * 1024)                                                                                           \
            (void)hipFuncSetAttribute((const void*)frontend_kernel<NBV, WRV, OCCV, S1V, MDV>,                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((frontend_kernel<NBV, WRV, OCCV, S1V, MDV>), dim3(grid), dim3(TPB), lds, st, a, tiles_per_utt,    \
                           n_tiles, seg, step, lo);                                                                    \
    } while (0)
#define AVSI_FE_LAUNCH(NBV)                   \
    do {                                      \
        if (model && occ == 3) AVSI_FE_LAUNCH1(NBV, 4, 3, true, 1); \
        else if (model) AVSI_FE_LAUNCH1(NBV, 4, 2, true, 1); \
        else if (small && step == 1) AVSI_FE_LAUNCH1(NBV, 4, 2, true, 0); \
        else if (small) AVSI_FE_LAUNCH1(NBV, 4, 2, false, 0); \
        else AVSI_FE_LAUNCH1(NBV, 8, 2, false, 0);      \
    } while (0)
    if (mode == 3) AVSI_FE_LAUNCH1(12, 4, 2, true, 3);
    else if (model_feat_only) AVSI_FE_LAUNCH1(12, 4, 2, true, 2);
    else if (nb == 8) AVSI_FE_LAUNCH(8);
    else if (nb == 12) AVSI_FE_LAUNCH(12);
    else AVSI_FE_LAUNCH(16);
#undef AVSI_FE_LAUNCH
#undef AVSI_FE_LAUNCH1
    return avsi_launch_status();
}

extern "C" int avsi_frontend_f32(const avsi_frontend_args* args, void* stream) {
    if (!args) return AVSI_ERR_INVALID_ARG;
    return frontend_launch(*args, 0, FeLoss{}, nullptr, stream);
}

extern "C" int avsi_frontend_l1_loss_supported(const avsi_frontend_args* args) {
    if (!args) return 0;
    const avsi_frontend_args& a = *args;
    const int nb_need = (a.frame_len + 31) / 32;
    return a.nfft == 512 && a.num_bins == 257 && a.mask && a.mean && a.stdev && a.spec_power == 1.f && a.log_spec && nb_need > 8 &&
           nb_need <= 12 && (int64_t)(FR - 1) * a.hop + 32 * 12 <= 4 * TPB * 4 && !(a.hop & 1) && !(a.frame_len & 1);
}

extern "C" int avsi_frontend_l1_loss_f32(const avsi_frontend_args* args, const float* pred, int64_t pred_stride_b,
                                         int64_t pred_stride_t, float* dpred, float grad_scale, float* out3, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    if (!args || !pred || !out3) return AVSI_ERR_INVALID_ARG;
    if (!avsi_frontend_l1_loss_supported(args)) return AVSI_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < (size_t)2 * AVSI_NUM_CU * 5 * sizeof(float)) return AVSI_ERR_WORKSPACE;
    avsi_frontend_args a = *args;
    a.out_stft = a.out_spec = a.out_feat = a.out_logmel = nullptr;      // the loss form stores nothing but dpred
    const FeLoss lo{pred, pred_stride_b, pred_stride_t, dpred, grad_scale, (float*)workspace};
    int grid = 0;
    const int rc = frontend_launch(a, 3, lo, &grid, stream);
    if (rc != AVSI_OK) return rc;
    return avsi_l1_final_launch((const float*)workspace, grid, (int64_t)a.batch * a.num_frames * 257, out3, (hipStream_t)stream);
}
