// Building blocks of the U-Net spectrogram inpainter (reference models.py:519-715 UNetFConvModel,
// unet_layers.py:6-37 encoder_layer_fconv / decoder_layer_fconv) on gfx950.
//
// Activations are NHWC, stored as 2-D [B*H*W rows][C] with row pitch ld (multiple of 4).  A
// convolution (tf.nn.conv2d, SAME, stride 1) is im2col + the fp32-MFMA GEMM of gemm.hip: the TF
// filter layout [kh][kw][Cin][Cout] IS the GEMM's B matrix with K ordered (kh, kw, c).  The im2col
// gather fuses what the decoder does in front of its convolution -- nearest-neighbour 2x
// up-sampling of the coarse input and the channel concat with the skip connection
// (tf.keras.layers.UpSampling2D + tf.concat, unet_layers.py:28-29) -- so neither is materialised.
//   avsi_im2col_f32      [B,H,W,C0] (+ up2x [B,H/2,W/2,C1]) -> col [B*H*W][Kc]
//   avsi_col2im_f32      gather-form adjoint (deterministic, no atomics), optional accumulate
//   avsi_colstats_f32    per-channel batch mean and 1/sqrt(var + eps)  (tf.layers.batch_normalization,
//                        training=True: batch statistics, biased variance, eps 1e-3)
//   avsi_bn_act_f32      y = act(gamma (x - mean) rstd + beta), act in {none, relu, leaky_relu(0.2)}
//   avsi_bn_act_bwd_f32  gradient of the above w.r.t. x, gamma, beta (two passes: sums, then apply)
//   avsi_maxpool2_f32 / avsi_maxpool2_bwd_f32   2x2 / stride 2 max pooling and its gradient
// All kernels are memory-bound grid-stride loops; the FLOPs live in the GEMMs.
#include "avsi_common.h"

namespace {

__device__ __attribute__((aligned(16))) float g_zero_pixel[64];      // zero-initialised: the out-of-image pixel
__device__ __forceinline__ float bn_act_one(float v, float sc, float sh, int act);

constexpr int TPB = 256;

inline int grid_for(int64_t items) {
    int64_t g = avsi_ceil_div(items, TPB);
    const int64_t cap = (int64_t)AVSI_NUM_CU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

struct ConvGeom {
    int B, H, W, C0, ld0, C1, ld1, k, Kc;
};

__global__ __launch_bounds__(TPB) void im2col_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                     float* __restrict__ col, const ConvGeom g) {
    const int Ct = g.C0 + g.C1, p = g.k / 2, H2 = g.H >> 1, W2 = g.W >> 1;
    const int64_t n = (int64_t)g.B * g.H * g.W * g.Kc;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int j = (int)(e % g.Kc);
        int64_t row = e / g.Kc;
        const int w = (int)(row % g.W);
        row /= g.W;
        const int h = (int)(row % g.H), b = (int)(row / g.H);
        float v = 0.f;
        if (j < g.k * g.k * Ct) {
            const int tap = j / Ct, c = j - tap * Ct;
            const int hh = h + tap / g.k - p, ww = w + tap % g.k - p;
            if (hh >= 0 && hh < g.H && ww >= 0 && ww < g.W) {
                if (c < g.C0)
                    v = s0[(((int64_t)b * g.H + hh) * g.W + ww) * g.ld0 + c];
                else
                    v = s1[(((int64_t)b * H2 + (hh >> 1)) * W2 + (ww >> 1)) * g.ld1 + (c - g.C0)];
            }
        }
        col[e] = v;
    }
}

// 16-byte form for layers whose channel counts are multiples of 4 (every layer but the three with a
// 1-channel operand): a group of 4 consecutive k never straddles a tap or the source boundary.
__global__ __launch_bounds__(TPB) void im2col4_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                      float* __restrict__ col, const ConvGeom g) {
    const int Ct = g.C0 + g.C1, p = g.k / 2, H2 = g.H >> 1, W2 = g.W >> 1, K4 = g.Kc >> 2;
    const int64_t n = (int64_t)g.B * g.H * g.W * K4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int j = (int)(e % K4) * 4;
        int64_t row = e / K4;
        const int w = (int)(row % g.W);
        row /= g.W;
        const int h = (int)(row % g.H), b = (int)(row / g.H);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < g.k * g.k * Ct) {
            const int tap = j / Ct, c = j - tap * Ct;
            const int hh = h + tap / g.k - p, ww = w + tap % g.k - p;
            if (hh >= 0 && hh < g.H && ww >= 0 && ww < g.W) {
                if (c < g.C0)
                    v = *reinterpret_cast<const float4*>(s0 + (((int64_t)b * g.H + hh) * g.W + ww) * g.ld0 + c);
                else
                    v = *reinterpret_cast<const float4*>(s1 + (((int64_t)b * H2 + (hh >> 1)) * W2 + (ww >> 1)) * g.ld1 + (c - g.C0));
            }
        }
        reinterpret_cast<float4*>(col)[e] = v;
    }
}

// Direct convolution for the three thin full-resolution layers of the U-Net (1 -> 16 channels 7x7,
// 1 + 16 -> 1 channels 3x3 over [skip, up2x(coarse)], 1 -> 1 channel 1x1): no MFMA shape fits a
// reduction of 49 / 153 / 1, and as im2col + GEMM they were 60 % of an inference step (the im2col
// matrix of the 17-channel layer alone is 5 GB at batch 512).  One thread per output pixel, lanes
// along W; the loops are fully unrolled, so the filter taps are wave-uniform scalar loads and every
// multiply-add has an SGPR operand.
// Direct convolution + ReLU + 2x2 max pooling in one pass (the first encoder layer at inference: 1 -> 16 channels
// 7 x 7, no batch norm, unet_layers.py:6-20 followed by the pooling): one thread per POOLED pixel computes the four
// outputs of its window and keeps their maximum, so the full-resolution activation (537 MB at batch 512, written
// and read back twice by the separate activation and pooling passes) never exists.
template <int K, int COUT>
__global__ __launch_bounds__(TPB) void direct_conv_relu_pool_kernel(const float* __restrict__ s0, int ld0,
                                                                    const float* __restrict__ filt, int ldf,
                                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                                    int ldo, int B, int H, int W) {
    constexpr int P = K / 2;
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t n = (int64_t)B * H2 * W2;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int w2 = (int)(e % W2);
        const int64_t bh = e / W2;
        const int h2 = (int)(bh % H2), b = (int)(bh / H2);
        // the (K + 1) x (K + 1) input patch of the 2 x 2 window, one channel
        float patch[K + 1][K + 1];
#pragma unroll
        for (int dh = 0; dh <= K; ++dh) {
            const int hh = 2 * h2 + dh - P;
#pragma unroll
            for (int dw = 0; dw <= K; ++dw) {
                const int ww = 2 * w2 + dw - P;
                const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
                patch[dh][dw] = s0[ok ? (((int64_t)b * H + hh) * W + ww) * ld0 : 0];
                patch[dh][dw] = ok ? patch[dh][dw] : 0.f;
            }
        }
        float best[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) best[o] = 0.f;          // ReLU: max(0, .)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                float acc[COUT];
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = bias ? bias[o] : 0.f;
#pragma unroll
                for (int dh = 0; dh < K; ++dh)
#pragma unroll
                    for (int dw = 0; dw < K; ++dw) {
                        const float xv = patch[py + dh][px + dw];
                        const float* wt = filt + (int64_t)(dh * K + dw) * ldf;
#pragma unroll
                        for (int o = 0; o < COUT; ++o) acc[o] += xv * wt[o];
                    }
#pragma unroll
                for (int o = 0; o < COUT; ++o) best[o] = fmaxf(best[o], acc[o]);
            }
        float* op = out + e * ldo;
#pragma unroll
        for (int o = 0; o < COUT; o += 4) *reinterpret_cast<float4*>(op + o) = make_float4(best[o], best[o + 1], best[o + 2], best[o + 3]);
    }
}

// The first encoder layer on the matrix cores: 7 x 7, one input channel -> 16, bias (+ ReLU + 2 x 2 max pooling).
// As a direct convolution it is VALU-bound (3136 multiply-adds per pooled pixel: 0.54 ms at batch 512, 24 TFLOP/s).
// Here a workgroup owns 16 x 64 output pixels of one image; their (16 + 6) x (64 + 6) input patch sits in LDS, and
// every group of 16 consecutive pixels of a row is a 16 x 49 by 49 x 16 product on v_mfma_f32_16x16x4_f32:
// 13 k-steps of four taps (49 padded to 52 with zero weights); lane l supplies A[pixel l % 16][tap 4 kk + l / 16] --
// one ds_read_b32 from the patch -- and B[tap][channel l % 16], thirteen registers loaded once per kernel.
// D[pixel 4 (l / 16) + i][channel l % 16], i = 0..3: horizontally adjacent pixels meet in one lane, vertically
// adjacent ones in the same lane of the next row's group, so ReLU + pooling are lane-local.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool POOL>
__global__ __launch_bounds__(256) void conv7_c1_mfma_kernel(const float* __restrict__ s0, int ld0, const float* __restrict__ filt,
                                                            int ldf, const float* __restrict__ bias, float* __restrict__ out,
                                                            int ldo, int H, int W) {
    constexpr int K = 7, P = 3, TH = 16, TW = 64, PW = TW + 2 * P + 2;      // patch row pitch 72
    __shared__ float patch[(TH + 2 * P) * PW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_w = W / TW, tiles_h = H / TH;
    const int tile = blockIdx.x % (tiles_w * tiles_h), b = blockIdx.x / (tiles_w * tiles_h);
    const int h0 = (tile / tiles_w) * TH, w0 = (tile % tiles_w) * TW;
    for (int i = tid; i < (TH + 2 * P) * (TW + 2 * P); i += 256) {
        const int r = i / (TW + 2 * P), c = i - r * (TW + 2 * P);
        const int hh = h0 + r - P, ww = w0 + c - P;
        const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
        const float v = s0[ok ? (((int64_t)b * H + hh) * W + ww) * ld0 : 0];
        patch[r * PW + c] = ok ? v : 0.f;
    }
    const int ch = lane & 15, kq = lane >> 4;
    float wreg[13];      // B operand: weight of tap 4 kk + kq for this lane's channel
    int aoff[13];        // patch offset of that tap relative to the pixel
#pragma unroll
    for (int kk = 0; kk < 13; ++kk) {
        const int t = 4 * kk + kq;
        wreg[kk] = t < K * K ? filt[(int64_t)t * ldf + ch] : 0.f;
        const int tt = t < K * K ? t : 0;
        aoff[kk] = (tt / K) * PW + (tt % K);
    }
    const float bv = bias ? bias[ch] : 0.f;
    __syncthreads();
    // wave wv owns rows 4 wv .. 4 wv + 3 of the tile, as two row pairs; four groups of 16 columns
    for (int rp = 0; rp < 2; ++rp) {
        const int r = 4 * wv + 2 * rp;
        for (int g = 0; g < TW / 16; ++g) {
            f32x4 d0 = {bv, bv, bv, bv}, d1 = {bv, bv, bv, bv};
            const float* p0 = patch + r * PW + 16 * g + (lane & 15);
            const float* p1 = p0 + PW;
#pragma unroll
            for (int kk = 0; kk < 13; ++kk) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0[aoff[kk]], wreg[kk], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1[aoff[kk]], wreg[kk], d1, 0, 0, 0);
            }
            if (POOL) {
                // pixels 4 kq + i of the group: pooled columns (16 g + 4 kq) / 2 + {0, 1}
                const float m0 = fmaxf(fmaxf(fmaxf(d0[0], d0[1]), fmaxf(d1[0], d1[1])), 0.f);
                const float m1 = fmaxf(fmaxf(fmaxf(d0[2], d0[3]), fmaxf(d1[2], d1[3])), 0.f);
                const int64_t prow = ((int64_t)b * (H / 2) + (h0 + r) / 2) * (W / 2) + (w0 + 16 * g + 4 * kq) / 2;
                out[prow * ldo + ch] = m0;
                out[(prow + 1) * ldo + ch] = m1;
            } else {
                const int64_t pix0 = ((int64_t)b * H + h0 + r) * W + w0 + 16 * g + 4 * kq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    out[(pix0 + i) * ldo + ch] = d0[i];
                    out[(pix0 + W + i) * ldo + ch] = d1[i];
                }
            }
        }
    }
}

// Convolutions with few channels at high resolution on v_mfma_f32_16x16x4_f32 (the decoder layer 16 + 32 -> 16 at
// 64 x 64 and the encoder layer 16 -> 32, 5 x 5, at 64 x 64): as implicit GEMMs on the 128 x 32 tile of gemm.hip they
// run at 41 / 83 TFLOP/s -- a 16-channel output fills half of every 32-column MFMA, and a wave's one 32 x 32 tile
// re-reads both operands from LDS for every MFMA.  Here a workgroup owns 4 x 32 output pixels; their input patch
// ((4 + KS - 1) x (32 + KS - 1) pixels, all CT channels of [skip, up2x(coarse)], channel pitch CT + 1: pixel-strided
// reads are conflict-free) and the whole filter sit in LDS; a wave owns one row = two groups of 16 pixels, and a
// group is a 16 x (KS^2 CT) by (KS^2 CT) x COUT product in k-steps of four channels of one tap: lane l supplies
// A[pixel l % 16][channel 4 kk' + l / 16] (one ds_read_b32 from the patch) and B[.][output channel l % 16] (one from
// the filter); D[pixel 4 (l / 16) + i][channel l % 16].
// STATS: the workgroup also leaves per-channel sums and sums of squares of ALL its output pixels in part[blk][2][COUT]
// (batch-norm statistics taken where the accumulators are; fixed reduction order: deterministic).
// BN1: s1 is the RAW convolution output of the layer below; its batch norm + LeakyReLU(0.2) (bn1 = mean, rstd, gamma, beta)
// are applied while the patch is staged -- that layer's own normalise / activate pass over its output does not exist.
struct Bn4 {
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
};
// PERSISTENT over tiles (round 4): a tile's MFMA work is ~3 us (216 - 400 MFMAs per wave) but staging its patch AND the whole
// filter (27 - 51 KB, the same for every tile) took 11 us more, one workgroup per tile.  Now a workgroup stages the filter
// ONCE, walks tiles blockIdx.x, + gridDim.x, ..., and fetches the NEXT tile's patch into registers (5 - 10 float4 per thread,
// branch-free) while the current tile's MFMAs run; statistics are carried in registers over the workgroup's tiles and
// reduced once at the end (gridDim.x partial rows instead of one per tile).
template <int KS, int C0, int C1, int COUT, bool STATS, bool BN1>
__global__ __launch_bounds__(256) void thin_mfma_conv_kernel(const float* __restrict__ s0, int ld0, const float* __restrict__ s1,
                                                             int ld1, const float* __restrict__ filt, int ldf,
                                                             const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                             int H, int W, const float* __restrict__ zeros,
                                                             float* __restrict__ part, const Bn4 bn1, const int n_tiles) {
    constexpr int P = KS / 2, CT = C0 + C1, CP = CT + 1, TH = 4, TW = 32, PH = TH + KS - 1, PWD = TW + KS - 1;
    constexpr int NCG = COUT / 16, CQ = CT / 4;
    constexpr int ITEMS = PH * PWD * CQ, NPRE = (ITEMS + 255) / 256;       // float4 items of a patch, per thread
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* patch = smem_f;                           // [PH][PWD][CP]
    float* wts = smem_f + PH * PWD * CP;             // [KS * KS * CT][COUT]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_w = W / TW, tiles_img = tiles_w * (H / TH);
    const int H2 = H >> 1, W2 = W >> 1;
    __shared__ float s_sc[BN1 ? C1 : 1], s_sh[BN1 ? C1 : 1];
    if (BN1 && tid < C1) {
        const float sc = bn1.gamma[tid] * bn1.rstd[tid];
        s_sc[tid] = sc, s_sh[tid] = bn1.beta[tid] - bn1.mean[tid] * sc;
    }
    // filter -> LDS (rows of COUT floats, 16-byte pieces), once per workgroup
    for (int i = tid; i < KS * KS * CT * (COUT / 4); i += 256) {
        const int row = i / (COUT / 4), q = i - row * (COUT / 4);
        *reinterpret_cast<float4*>(wts + row * COUT + 4 * q) = *reinterpret_cast<const float4*>(filt + (int64_t)row * ldf + 4 * q);
    }
    // a tile's patch: one float4 (four channels of one pixel) per item; pixels outside the image read a page of zeros
    float4 pre[NPRE];
    unsigned okbits = 0;
    auto fetch = [&](int tile) {
        const int b = tile / tiles_img, tl = tile - b * tiles_img;
        const int h0 = (tl / tiles_w) * TH, w0 = (tl - (tl / tiles_w) * tiles_w) * TW;
        okbits = 0;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + 256 * j;
            const int ic = i < ITEMS ? i : 0;                     // (threads past the last item re-read item 0: never stored)
            const int pix = ic / CQ, c4 = (ic - pix * CQ) * 4;
            const int pr = pix / PWD, pc = pix - pr * PWD;
            const int hh = h0 + pr - P, ww = w0 + pc - P;
            const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
            const float* src = zeros;
            if (ok) {
                if (c4 < C0) src = s0 + (((int64_t)b * H + hh) * W + ww) * ld0 + c4;
                else src = s1 + (((int64_t)b * H2 + (hh >> 1)) * W2 + (ww >> 1)) * ld1 + (c4 - C0);
            }
            pre[j] = *reinterpret_cast<const float4*>(src);
            okbits |= (unsigned)ok << j;
        }
    };
    const int px = lane & 15, kq = lane >> 4;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    float bv[NCG], st1[NCG], st2[NCG];
#pragma unroll
    for (int n = 0; n < NCG; ++n) bv[n] = bias ? bias[16 * n + px] : 0.f, st1[n] = 0.f, st2[n] = 0.f;
    const float* prow = patch + (wv * PWD + px) * CP + kq;       // row wv of the tile, tap (0, 0), this lane's channel
    const float* wlan = wts + kq * COUT + px;
    if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
    __syncthreads();                                             // filter and batch-norm constants in LDS
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // ---- the patch this thread fetched -> LDS (the batch norm + activation of a raw source applied on the way)
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + 256 * j;
            if (i < ITEMS) {
                const int pix = i / CQ, c4 = (i - pix * CQ) * 4;
                float4 v = pre[j];
                if (BN1 && ((okbits >> j) & 1u) && c4 >= C0) {
                    const int c = c4 - C0;
                    v.x = bn_act_one(v.x, s_sc[c], s_sh[c], 2), v.y = bn_act_one(v.y, s_sc[c + 1], s_sh[c + 1], 2);
                    v.z = bn_act_one(v.z, s_sc[c + 2], s_sh[c + 2], 2), v.w = bn_act_one(v.w, s_sc[c + 3], s_sh[c + 3], 2);
                }
                float* d = patch + pix * CP + c4;
                d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
            }
        }
        AVSI_LDS_BARRIER();
        if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);      // in flight during the MFMAs below
        const int b = tile / tiles_img, tl = tile - b * tiles_img;
        const int h0 = (tl / tiles_w) * TH, w0 = (tl - (tl / tiles_w) * tiles_w) * TW;
        f32x4v acc[2][NCG];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int n = 0; n < NCG; ++n) acc[g][n] = (f32x4v){bv[n], bv[n], bv[n], bv[n]};
#pragma unroll 1
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int dh = tap / KS, dw = tap - dh * KS;
            const float* pa = prow + (dh * PWD + dw) * CP;
            const float* pw = wlan + tap * CT * COUT;
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                const float a0 = pa[4 * cq], a1 = pa[16 * CP + 4 * cq];          // the wave's two pixel groups
#pragma unroll
                for (int n = 0; n < NCG; ++n) {
                    const float wv_ = pw[4 * cq * COUT + 16 * n];
                    acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, wv_, acc[0][n], 0, 0, 0);
                    acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, wv_, acc[1][n], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int64_t pix0 = ((int64_t)b * H + h0 + wv) * W + w0 + 16 * g + 4 * kq;
#pragma unroll
            for (int n = 0; n < NCG; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[g][n][i];
                    out[(pix0 + i) * ldo + 16 * n + px] = v;
                    if (STATS) st1[n] += v, st2[n] += v * v;
                }
        }
        AVSI_LDS_BARRIER();                              // every wave is done reading the patch
    }
    if (STATS) {
        // lane (px, kq) holds channel 16 n + px of its pixels of every tile: the four kq groups (shuffles), then the four waves
        float* red = smem_f;                             // [4 waves][2][COUT] (the patch is dead)
#pragma unroll
        for (int n = 0; n < NCG; ++n) {
            float t1 = st1[n], t2 = st2[n];
            t1 += __shfl_xor(t1, 16, 64), t2 += __shfl_xor(t2, 16, 64);
            t1 += __shfl_xor(t1, 32, 64), t2 += __shfl_xor(t2, 32, 64);
            if (kq == 0) red[(wv * 2 + 0) * COUT + 16 * n + px] = t1, red[(wv * 2 + 1) * COUT + 16 * n + px] = t2;
        }
        __syncthreads();
        if (tid < 2 * COUT) {
            const float v = (red[tid] + red[2 * COUT + tid]) + (red[4 * COUT + tid] + red[6 * COUT + tid]);
            part[(int64_t)blockIdx.x * 2 * COUT + tid] = v;
        }
    }
}

// Filter gradient of the few-channel layers on the 16-wide MFMA (round 5): dW[(tap, c)][n] = sum over pixels of
// in(p + off(tap), c) dY[p][n].  As a split-K implicit GEMM (gemm_dma_kernel<true, ..., 32>) the three thin layers gathered
// every input value once PER TAP through the DMA -- 3.6 GB for 29 GFLOP at the 16 + 32 -> 16 layer -- and ran at 30 - 57
// TFLOP/s: 2.8 of 17 ms per training step at 512 clips.  Here the tile's input patch goes through LDS once (the staging of
// thin_mfma_conv_kernel: same 4 x 32-pixel tiles, same patch, the next tile's prefetched into registers) next to the
// tile's dY; the (tap, 16-channel group) pairs are dealt to the four waves, and a pair's 16 x COUT block of dW is
// v_mfma_f32_16x16x4_f32 accumulators that live in registers over ALL the tiles of the (persistent) workgroup:
//   A[row = channel l % 16][k = pixel l / 16]  -- one ds_read_b32 from the patch at the tap's offset,
//   B[k = pixel l / 16][col = output channel l % 16] -- one from the dY tile, shared by all pairs of the wave,
//   D[row = channel 4 (l / 16) + i][col = output channel l % 16].
// One partial filter per workgroup in `part` [gridDim.x][KS^2 CT][COUT]; the caller sums them in order (deterministic).
template <int KS, int C0, int C1, int COUT>
__global__ __launch_bounds__(256) void thin_mfma_wgrad_kernel(const float* __restrict__ s0, int ld0, const float* __restrict__ s1,
                                                              int ld1, const float* __restrict__ dy, int ldy,
                                                              float* __restrict__ part, int H, int W,
                                                              const float* __restrict__ zeros, const int n_tiles) {
    constexpr int P = KS / 2, CT = C0 + C1, CP = CT + 1, TH = 4, TW = 32, PH = TH + KS - 1, PWD = TW + KS - 1;
    constexpr int NCG = COUT / 16, MCG = CT / 16, CQ = CT / 4, DP = COUT + 1;
    constexpr int ITEMS = PH * PWD * CQ, NPRE = (ITEMS + 255) / 256;       // float4 items of a patch, per thread
    constexpr int DITEMS = TH * TW * (COUT / 4), NDY = DITEMS / 256;       // float4 items of the dY tile, per thread
    constexpr int NT = KS * KS * MCG, TPW = (NT + 3) / 4;                  // (tap, channel group) pairs, per wave
    static_assert(DITEMS % 256 == 0 && CT % 16 == 0 && COUT % 16 == 0, "tile shapes");
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* patch = smem_f;                           // [PH][PWD][CP]
    float* dyt = smem_f + PH * PWD * CP;             // [TH * TW][DP]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_w = W / TW, tiles_img = tiles_w * (H / TH);
    const int H2 = H >> 1, W2 = W >> 1;
    float4 pre[NPRE], dpre[NDY];
    auto fetch = [&](int tile) {
        const int b = tile / tiles_img, tl = tile - b * tiles_img;
        const int h0 = (tl / tiles_w) * TH, w0 = (tl - (tl / tiles_w) * tiles_w) * TW;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + 256 * j;
            const int ic = i < ITEMS ? i : 0;                     // (threads past the last item re-read item 0: never stored)
            const int pix = ic / CQ, c4 = (ic - pix * CQ) * 4;
            const int pr = pix / PWD, pc = pix - pr * PWD;
            const int hh = h0 + pr - P, ww = w0 + pc - P;
            const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
            const float* src = zeros;                             // pixels outside the image read a page of zeros
            if (ok) {
                if (c4 < C0) src = s0 + (((int64_t)b * H + hh) * W + ww) * ld0 + c4;
                else src = s1 + (((int64_t)b * H2 + (hh >> 1)) * W2 + (ww >> 1)) * ld1 + (c4 - C0);
            }
            pre[j] = *reinterpret_cast<const float4*>(src);
        }
#pragma unroll
        for (int j = 0; j < NDY; ++j) {
            const int i = tid + 256 * j;
            const int pix = i / (COUT / 4), q = i - pix * (COUT / 4);
            const int r = pix / TW, c = pix - r * TW;
            dpre[j] = *reinterpret_cast<const float4*>(dy + (((int64_t)b * H + h0 + r) * W + w0 + c) * ldy + 4 * q);
        }
    };
    const int px = lane & 15, kq = lane >> 4;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v acc[TPW][NCG];
    int aoff[TPW];                                   // patch offset of pair j of this wave: its tap and channel group
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = wv * TPW + j, tc = t < NT ? t : 0;
        const int tap = tc / MCG, cg = tc - tap * MCG;
        const int dh = tap / KS, dw = tap - dh * KS;
        aoff[j] = (dh * PWD + dw) * CP + 16 * cg;
#pragma unroll
        for (int n = 0; n < NCG; ++n) acc[j][n] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    }
    if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + 256 * j;
            if (i < ITEMS) {
                const int pix = i / CQ, c4 = (i - pix * CQ) * 4;
                float* d = patch + pix * CP + c4;
                d[0] = pre[j].x, d[1] = pre[j].y, d[2] = pre[j].z, d[3] = pre[j].w;
            }
        }
#pragma unroll
        for (int j = 0; j < NDY; ++j) {
            const int i = tid + 256 * j;
            const int pix = i / (COUT / 4), q = i - pix * (COUT / 4);
            float* d = dyt + pix * DP + 4 * q;
            d[0] = dpre[j].x, d[1] = dpre[j].y, d[2] = dpre[j].z, d[3] = dpre[j].w;
        }
        AVSI_LDS_BARRIER();
        if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);      // in flight during the MFMAs below
        // Straight-line over the 32 k-steps of the tile, the fragments of step kk + 1 read while the MFMAs of step kk run
        // (first version: a branch, a read and s_waitcnt lgkmcnt(0) in front of EVERY MFMA -- 35 TFLOP/s).  A wave whose
        // last slot has no pair (NT is no multiple of 4) accumulates pair 0 once more there and never stores it.
        const float* pa0 = patch + kq * CP + px;
        const float* pb0 = dyt + kq * DP + px;
        float af[2][TPW], bf[2][NCG];
#pragma unroll
        for (int n = 0; n < NCG; ++n) bf[0][n] = pb0[16 * n];
#pragma unroll
        for (int j = 0; j < TPW; ++j) af[0][j] = pa0[aoff[j]];
#pragma unroll
        for (int kk = 0; kk < TH * TW / 4; ++kk) {
            constexpr int KSTEPS = TH * TW / 4;
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < KSTEPS) {
                const int r = (kk + 1) >> 3, c = ((kk + 1) & 7) << 2;      // compile-time: the loop is unrolled
#pragma unroll
                for (int n = 0; n < NCG; ++n) bf[nxt][n] = pb0[(r * TW + c) * DP + 16 * n];
#pragma unroll
                for (int j = 0; j < TPW; ++j) af[nxt][j] = pa0[(r * PWD + c) * CP + aoff[j]];
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int n = 0; n < NCG; ++n)
                    acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][j], bf[cur][n], acc[j][n], 0, 0, 0);
        }
        AVSI_LDS_BARRIER();                              // every wave is done reading the patch and the dY tile
    }
    float* prow = part + (int64_t)blockIdx.x * (KS * KS * CT * COUT);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = wv * TPW + j;
        if (t < NT) {
            const int tap = t / MCG, cg = t - tap * MCG;
#pragma unroll
            for (int n = 0; n < NCG; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) prow[(tap * CT + 16 * cg + 4 * kq + i) * COUT + 16 * n + px] = acc[j][n][i];
        }
    }
}

// Filter gradient of the FIRST layer (7 x 7, one input channel -> 16) on the 16-wide MFMA: with one channel the M side of the
// product is the TAPS -- dW[tap][n] = sum_p x(p + off(tap)) dY[p][n], 49 taps in four groups of 16 (one per wave; the 15 slots
// past tap 48 multiply by a zero weight and are not stored).  A[row = tap][k = pixel]: every lane gathers its own tap's
// offset from the 14 x 38 one-channel patch of an 8 x 32-pixel tile; B = the dY tile as in thin_mfma_wgrad_kernel.  The
// direct form (thin_wgrad_kernel<7, 1, 0, 16>: 784 accumulators per thread, seven workgroup rows) ran at 17 TFLOP/s, 0.78 ms.
__global__ __launch_bounds__(256) void conv7_c1_wgrad_mfma_kernel(const float* __restrict__ s0, int ld0, const float* __restrict__ dy,
                                                                 int ldy, float* __restrict__ part, int H, int W,
                                                                 const int n_tiles) {
    constexpr int KS = 7, P = 3, TH = 8, TW = 32, PH = TH + KS - 1, PWD = TW + KS - 1, COUT = 16, DP = COUT + 1;
    constexpr int PITEMS = PH * PWD, NPRE = (PITEMS + 255) / 256, NDY = TH * TW * (COUT / 4) / 256, KSTEPS = TH * TW / 4;
    __shared__ float patch[PH * PWD + 8];
    __shared__ float dyt[TH * TW * DP];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_w = W / TW, tiles_img = tiles_w * (H / TH);
    float pre[NPRE];
    float4 dpre[NDY];
    auto fetch = [&](int tile) {
        const int b = tile / tiles_img, tl = tile - b * tiles_img;
        const int h0 = (tl / tiles_w) * TH, w0 = (tl - (tl / tiles_w) * tiles_w) * TW;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + 256 * j;
            const int ic = i < PITEMS ? i : 0;
            const int pr = ic / PWD, pc = ic - pr * PWD;
            const int hh = h0 + pr - P, ww = w0 + pc - P;
            const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
            pre[j] = *(ok ? s0 + (((int64_t)b * H + hh) * W + ww) * ld0 : g_zero_pixel);       // unconditional loads
        }
#pragma unroll
        for (int j = 0; j < NDY; ++j) {
            const int i = tid + 256 * j;
            const int pix = i >> 2, q = i & 3;
            const int r = pix / TW, c = pix - r * TW;
            dpre[j] = *reinterpret_cast<const float4*>(dy + (((int64_t)b * H + h0 + r) * W + w0 + c) * ldy + 4 * q);
        }
    };
    const int px = lane & 15, kq = lane >> 4;
    const int tap = 16 * wv + px;
    const bool live = tap < KS * KS;
    const int tc = live ? tap : 0;
    const int toff = (tc / KS) * PWD + (tc % KS) + kq;           // this lane's tap offset + its pixel of a k-step
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + 256 * j;
            if (i < PITEMS) patch[i] = pre[j];
        }
#pragma unroll
        for (int j = 0; j < NDY; ++j) {
            const int i = tid + 256 * j;
            float* d = dyt + (i >> 2) * DP + 4 * (i & 3);
            d[0] = dpre[j].x, d[1] = dpre[j].y, d[2] = dpre[j].z, d[3] = dpre[j].w;
        }
        AVSI_LDS_BARRIER();
        if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
        const float* pa = patch + toff;
        const float* pb = dyt + kq * DP + px;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int r = kk >> 3, c = (kk & 7) << 2;
            const float a = live ? pa[r * PWD + c] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pb[(r * TW + c) * DP], acc, 0, 0, 0);
        }
        AVSI_LDS_BARRIER();
    }
    float* prow = part + (int64_t)blockIdx.x * (KS * KS * COUT);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = 16 * wv + 4 * kq + i;
        if (t < KS * KS) prow[t * COUT + px] = acc[i];
    }
}

// Filter gradient of the LAST 3 x 3 layer (1 skip channel ++ 16 up-sampled channels -> 1) on the 16-wide MFMA.  One output
// channel would leave 15 of 16 MFMA columns empty, so the sum is turned around: dW[tap][c] = sum_q in(q, c) dY(q - off(tap)) --
// the M side are the 16 coarse channels of input pixel q, the N side the nine taps (a per-lane gather from the one-channel dY
// patch, columns 9 .. 15 multiply by zero), K the pixels.  The skip channel's nine sums are plain multiply-adds (one pixel per
// thread).  8 x 32-pixel tiles, persistent workgroups, the four waves take 16 of a tile's 64 k-steps each.
// The direct form (thin_wgrad_kernel<3, 1, 16, 1>) took 0.33 - 0.58 ms at 512 clips.
__global__ __launch_bounds__(256) void conv3_c17_wgrad_mfma_kernel(const float* __restrict__ s0, int ld0, const float* __restrict__ s1,
                                                                  int ld1, const float* __restrict__ dy, int ldy,
                                                                  float* __restrict__ part, int H, int W, const int n_tiles) {
    constexpr int TH = 8, TW = 32, CH = TH / 2, CW = TW / 2, CP = 17, DH = TH + 2, DW = TW + 2, KSTEPS = TH * TW / 4;
    __shared__ float coarse[CH * CW * CP];
    __shared__ float dyp[DH * DW + 6];
    __shared__ float fine[TH * TW];
    __shared__ float red[4][16 * 16 + 16];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_w = W / TW, tiles_img = tiles_w * (H / TH);
    const int H2 = H >> 1, W2 = W >> 1;
    float4 cpre;
    float dpre[2], fpre;
    auto fetch = [&](int tile) {
        const int b = tile / tiles_img, tl = tile - b * tiles_img;
        const int h0 = (tl / tiles_w) * TH, w0 = (tl - (tl / tiles_w) * tiles_w) * TW;
        {   // coarse pixels of the tile: 4 x 16, four float4 each
            const int pix = tid >> 2, q = tid & 3;
            const int r = pix / CW, c = pix - r * CW;
            cpre = *reinterpret_cast<const float4*>(s1 + (((int64_t)b * H2 + (h0 >> 1) + r) * W2 + (w0 >> 1) + c) * ld1 + 4 * q);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j;
            const int ic = i < DH * DW ? i : 0;
            const int r = ic / DW, c = ic - r * DW;
            const int hh = h0 + r - 1, ww = w0 + c - 1;
            const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
            dpre[j] = *(ok ? dy + (((int64_t)b * H + hh) * W + ww) * ldy : g_zero_pixel);
        }
        fpre = s0[(((int64_t)b * H + h0 + (tid >> 5)) * W + w0 + (tid & 31)) * ld0];
    };
    const int px = lane & 15, kq = lane >> 4;
    const bool tap_live = px < 9;
    const int tp = tap_live ? px : 0;
    const int boff = (2 - tp / 3) * DW + (2 - tp % 3) + kq;      // dY(q - off(tap)) in the patch (origin -1, -1), this lane's pixel
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    float facc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) facc[t] = 0.f;
    if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        {
            float* d = coarse + (tid >> 2) * CP + 4 * (tid & 3);
            d[0] = cpre.x, d[1] = cpre.y, d[2] = cpre.z, d[3] = cpre.w;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (tid + 256 * j < DH * DW) dyp[tid + 256 * j] = dpre[j];
        fine[tid] = fpre;
        AVSI_LDS_BARRIER();
        if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
#pragma unroll
        for (int k = 0; k < KSTEPS / 4; ++k) {
            const int kk = wv * (KSTEPS / 4) + k;
            const int r = kk >> 3, c = ((kk & 7) << 2) + kq;
            const float a = coarse[((r >> 1) * CW + (c >> 1)) * CP + px];
            const float bq = dyp[r * DW + ((kk & 7) << 2) + boff];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, tap_live ? bq : 0.f, acc, 0, 0, 0);
        }
        {   // the skip channel: pixel (tid / 32, tid % 32) of the tile
            const int r = tid >> 5, c = tid & 31;
            const float x = fine[tid];
#pragma unroll
            for (int t = 0; t < 9; ++t) facc[t] += x * dyp[(r + 2 - t / 3) * DW + c + 2 - t % 3];
        }
        AVSI_LDS_BARRIER();
    }
    // ---- the workgroup's partial filter: waves added in order, then [tap][17]
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wv][(4 * kq + i) * 16 + px] = acc[i];            // [channel][tap]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v = facc[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wv][256 + t] = v;
    }
    __syncthreads();
    float* prow = part + (int64_t)blockIdx.x * 153;
    if (tid < 153) {
        const int tap = tid / 17, c = tid - tap * 17;
        const int src = c == 0 ? 256 + tap : (c - 1) * 16 + tap;
        prow[tid] = (red[0][src] + red[1][src]) + (red[2][src] + red[3][src]);
    }
}

template <int K, int C0, int C1, int COUT>
__global__ __launch_bounds__(TPB) void direct_conv_kernel(const float* __restrict__ s0, int ld0, const float* __restrict__ s1,
                                                          int ld1, const float* __restrict__ filt, int ldf,
                                                          const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                          int B, int H, int W) {
    constexpr int P = K / 2, CT = C0 + C1;
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t n = (int64_t)B * H * W;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int w = (int)(e % W);
        const int64_t bh = e / W;
        const int h = (int)(bh % H), b = (int)(bh / H);
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = bias ? bias[o] : 0.f;
#pragma unroll
        for (int dh = 0; dh < K; ++dh) {
            const int hh = h + dh - P;
#pragma unroll
            for (int dw = 0; dw < K; ++dw) {
                const int ww = w + dw - P;
                const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
                const float* wt = filt + (int64_t)((dh * K + dw) * CT) * ldf;
                // pixels outside the image read a page of zeros: the loads stay unconditional (a load under a per-lane
                // test is closed with s_waitcnt vmcnt(0) right behind it, and the 36 + 9 loads of a 3 x 3 x 17 pixel
                // then wait for each other in turn: 0.31 ms for the 17 -> 1 layer at batch 512)
                if (C0 > 0) {
                    const float* px = ok ? s0 + (((int64_t)b * H + hh) * W + ww) * ld0 : g_zero_pixel;
#pragma unroll
                    for (int c = 0; c < C0; ++c) {
                        const float xv = px[c];
#pragma unroll
                        for (int o = 0; o < COUT; ++o) acc[o] += xv * wt[c * ldf + o];
                    }
                }
                if (C1 > 0) {
                    const float* px = ok ? s1 + (((int64_t)b * H2 + (hh >> 1)) * W2 + (ww >> 1)) * ld1 : g_zero_pixel;
#pragma unroll
                    for (int c4 = 0; c4 < C1; c4 += 4) {
                        const float4 xv = *reinterpret_cast<const float4*>(px + c4);
                        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int o = 0; o < COUT; ++o) acc[o] += xs[q] * wt[(C0 + c4 + q) * ldf + o];
                    }
                }
            }
        }
        float* op = out + e * ldo;
        if (COUT % 4 == 0) {
#pragma unroll
            for (int o = 0; o < COUT; o += 4) *reinterpret_cast<float4*>(op + o) = make_float4(acc[o], acc[o + 1], acc[o + 2], acc[o + 3]);
        } else {
#pragma unroll
            for (int o = 0; o < COUT; ++o) op[o] = acc[o];
        }
    }
}

// The 3 x 3 output layer over (1 skip channel ++ 16 up-sampled channels) -> 1, tiled: an 8 x 32-pixel tile's input -- the
// fine channel and the coarse (half-resolution) 16-channel pixels it up-samples from -- and the 153 weights go through
// LDS once, then every thread owns one pixel: 9 taps x (1 + 4 x 16-byte) LDS reads.  The per-pixel form above fetched
// every coarse pixel 36 times through the L1 (0.31 ms at batch 512 for 2.6 GFLOP).
// STATS: the block also leaves the sum and the sum of squares of its 256 outputs in part[2 blk], part[2 blk + 1] (fixed
// reduction tree: deterministic) -- the batch statistics of the layer's one channel without a pass over its output.
template <bool STATS, bool BN1>
__global__ __launch_bounds__(256) void conv3_c17_out1_kernel(const float* __restrict__ s0, int ld0, const float* __restrict__ s1,
                                                             int ld1, const float* __restrict__ filt, int ldf,
                                                             const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                             int H, int W, float* __restrict__ part, const Bn4 bn1) {
    constexpr int TH = 8, TW = 32, FH = TH + 2, FW = TW + 2, CH = TH / 2 + 2, CW = TW / 2 + 2, CP = 20, WP = 20;
    __shared__ float fine[FH * FW];
    __shared__ __attribute__((aligned(16))) float coarse[CH * CW * CP];     // pitch 20: conflict-free 16-byte reads
    __shared__ __attribute__((aligned(16))) float wts[9 * WP];             // [tap][0] fine weight, [tap][4 .. 19] coarse
    const int tid = threadIdx.x;
    const int tiles_w = W / TW, tiles_h = H / TH;
    const int tile = blockIdx.x % (tiles_w * tiles_h), b = blockIdx.x / (tiles_w * tiles_h);
    const int h0 = (tile / tiles_w) * TH, w0 = (tile % tiles_w) * TW;
    const int H2 = H >> 1, W2 = W >> 1;
    const int ch0 = (h0 >> 1) - 1, cw0 = (w0 >> 1) - 1;                    // h0, w0 are even
    if (tid < 9 * 17) {
        const int tap = tid / 17, c = tid - tap * 17;
        wts[tap * WP + (c == 0 ? 0 : 3 + c)] = filt[(int64_t)(tap * 17 + c) * ldf];
    }
    __shared__ float s_sc[16], s_sh[16];
    if (BN1 && tid >= 192 && tid < 208) {       // (the raw 16-channel output of the layer below: see thin_mfma_conv_kernel)
        const int c = tid - 192;
        const float sc = bn1.gamma[c] * bn1.rstd[c];
        s_sc[c] = sc, s_sh[c] = bn1.beta[c] - bn1.mean[c] * sc;
    }
    if (BN1) __syncthreads();
    for (int i = tid; i < FH * FW; i += 256) {
        const int r = i / FW, c = i - r * FW;
        const int hh = h0 + r - 1, ww = w0 + c - 1;
        const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
        fine[i] = *(ok ? s0 + (((int64_t)b * H + hh) * W + ww) * ld0 : g_zero_pixel);
    }
    for (int i = tid; i < CH * CW * 4; i += 256) {
        const int pix = i >> 2, q = i & 3;
        const int r = pix / CW, c = pix - r * CW;
        const int hh = ch0 + r, ww = cw0 + c;
        const bool ok = hh >= 0 && hh < H2 && ww >= 0 && ww < W2;
        float4 v = *reinterpret_cast<const float4*>(ok ? s1 + (((int64_t)b * H2 + hh) * W2 + ww) * ld1 + 4 * q : g_zero_pixel);
        if (BN1 && ok) {
            const int c = 4 * q;
            v.x = bn_act_one(v.x, s_sc[c], s_sh[c], 2), v.y = bn_act_one(v.y, s_sc[c + 1], s_sh[c + 1], 2);
            v.z = bn_act_one(v.z, s_sc[c + 2], s_sh[c + 2], 2), v.w = bn_act_one(v.w, s_sc[c + 3], s_sh[c + 3], 2);
        }
        *reinterpret_cast<float4*>(coarse + pix * CP + 4 * q) = v;
    }
    __syncthreads();
    const int r = tid >> 5, c = tid & 31;
    float acc = bias ? bias[0] : 0.f;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            const float* wt = wts + (dh * 3 + dw) * WP;
            acc += fine[(r + dh) * FW + c + dw] * wt[0];
            // fine pixel (h0 + r + dh - 1, w0 + c + dw - 1) up-samples coarse pixel (that >> 1); a fine pixel outside
            // the image has its coarse pixel outside too (H, W even), i.e. zeros in the patch
            const int cr = ((h0 + r + dh - 1) >> 1) - ch0, cc = ((w0 + c + dw - 1) >> 1) - cw0;
            const float* cp = coarse + (cr * CW + cc) * CP;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x = *reinterpret_cast<const float4*>(cp + 4 * q);
                const float4 w = *reinterpret_cast<const float4*>(wt + 4 + 4 * q);
                acc += x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w;
            }
        }
    out[(((int64_t)b * H + h0 + r) * W + w0 + c) * ldo] = acc;
    if (STATS) {
        __shared__ float red[2][4];
        float t1 = acc, t2 = acc * acc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t1 += __shfl_xor(t1, o, 64), t2 += __shfl_xor(t2, o, 64);
        if ((tid & 63) == 0) red[0][tid >> 6] = t1, red[1][tid >> 6] = t2;
        __syncthreads();
        if (tid == 0) {
            part[2 * (int64_t)blockIdx.x + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            part[2 * (int64_t)blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}

// The rest of the network behind that convolution, one pass (inference form): batch norm of the one channel, LeakyReLU(0.2),
// the 1 x 1 output convolution (a scalar weight and bias: models.py:607) and the sequence mask of `prediction` (models.py:609-615).
__global__ __launch_bounds__(TPB) void unet_tail_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ w_out,
                                                        const float* __restrict__ b_out, const long long* __restrict__ seq_len,
                                                        float* __restrict__ logits, float* __restrict__ pred, int64_t n4, int T,
                                                        int F) {
    const float sc = gamma[0] * rstd[0], sh = beta[0] - mean[0] * sc, w = w_out[0], bo = b_out ? b_out[0] : 0.f;
    const int64_t tf4 = (int64_t)T * F / 4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n4; e += (int64_t)gridDim.x * TPB) {
        const float4 v = reinterpret_cast<const float4*>(x)[e];
        const int64_t b = e / tf4;
        const int t = (int)((e - b * tf4) * 4 / F);              // F is a multiple of 4: the four values share a frame
        const float m = t < seq_len[b] ? 1.f : 0.f;
        float4 y;
        y.x = w * bn_act_one(v.x, sc, sh, 2) + bo, y.y = w * bn_act_one(v.y, sc, sh, 2) + bo;
        y.z = w * bn_act_one(v.z, sc, sh, 2) + bo, y.w = w * bn_act_one(v.w, sc, sh, 2) + bo;
        if (logits) reinterpret_cast<float4*>(logits)[e] = y;
        reinterpret_cast<float4*>(pred)[e] = make_float4(y.x * m, y.y * m, y.z * m, y.w * m);
    }
}

// Filter gradient of the thin layers, direct form: dW[(tap, c)][n] = sum over pixels of in(p + off(tap), c) dY[p][n].
// blockIdx.y = filter row dh (K rows), so a thread keeps K * CT * COUT accumulators; pixels are
// grid-strided, lanes along W.  Per block: wave shuffle tree, LDS across the 4 waves, one partial
// row per block in `part` [gridDim.x][K*K*CT*COUT]; the caller sums the rows in order (deterministic).
template <int K, int C0, int C1, int COUT>
__global__ __launch_bounds__(TPB) void thin_wgrad_kernel(const float* __restrict__ s0, int ld0, const float* __restrict__ s1,
                                                         int ld1, const float* __restrict__ dy, int ldy,
                                                         float* __restrict__ part, int B, int H, int W) {
    constexpr int P = K / 2, CT = C0 + C1, NA = K * CT * COUT;
    __shared__ float red[TPB / 64][NA];
    const int dh = blockIdx.y;
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t n = (int64_t)B * H * W;
    float acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int w = (int)(e % W);
        const int64_t bh = e / W;
        const int h = (int)(bh % H), b = (int)(bh / H);
        float g[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) g[o] = dy[e * ldy + o];
        const int hh = h + dh - P;
        if (hh < 0 || hh >= H) continue;
#pragma unroll
        for (int dw = 0; dw < K; ++dw) {
            const int ww = w + dw - P;
            const bool ok = ww >= 0 && ww < W;
            if (C0 > 0) {
                const float* px = ok ? s0 + (((int64_t)b * H + hh) * W + ww) * ld0 : g_zero_pixel;      // unconditional loads
#pragma unroll
                for (int c = 0; c < C0; ++c) {
                    const float xv = px[c];
#pragma unroll
                    for (int o = 0; o < COUT; ++o) acc[(dw * CT + c) * COUT + o] += xv * g[o];
                }
            }
            if (C1 > 0) {
                const float* px = ok ? s1 + (((int64_t)b * H2 + (hh >> 1)) * W2 + (ww >> 1)) * ld1 : g_zero_pixel;
#pragma unroll
                for (int c4 = 0; c4 < C1; c4 += 4) {
                    const float4 xv = *reinterpret_cast<const float4*>(px + c4);
                    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int o = 0; o < COUT; ++o) acc[(dw * CT + C0 + c4 + q) * COUT + o] += xs[q] * g[o];
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NA; i += TPB) {
        float v = 0.f;
        for (int q = 0; q < TPB / 64; ++q) v += red[q][i];
        part[(int64_t)blockIdx.x * (K * NA) + dh * NA + i] = v;
    }
}

// Input gradient of the 1 + 16 -> 1 layer w.r.t. its up-sampled source: every coarse pixel collects the
// 2x2 fine pixels it was copied to, each through the 9 taps (adjoint of up2x followed by the 3x3 conv).
__global__ __launch_bounds__(TPB) void thin_dx_coarse_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ filt,
                                                             int ldf, float* __restrict__ d1, int ld1, int accumulate, int B,
                                                             int H, int W) {
    constexpr int K = 3, C0 = 1, C1 = 16, CT = 17;
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t n = (int64_t)B * H2 * W2;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int w2 = (int)(e % W2);
        const int64_t bh = e / W2;
        const int h2 = (int)(bh % H2), b = (int)(bh / H2);
        float acc[C1];
#pragma unroll
        for (int c = 0; c < C1; ++c) acc[c] = 0.f;
#pragma unroll
        for (int fh = 0; fh < 2; ++fh)
#pragma unroll
            for (int fw = 0; fw < 2; ++fw)
#pragma unroll
                for (int dh = 0; dh < K; ++dh)
#pragma unroll
                    for (int dw = 0; dw < K; ++dw) {
                        // fine pixel (2 h2 + fh, 2 w2 + fw) was read by output pixel p = fine - off(tap)
                        const int ph = 2 * h2 + fh - (dh - 1), pw = 2 * w2 + fw - (dw - 1);
                        const float g = *((ph >= 0 && ph < H && pw >= 0 && pw < W) ? dy + (((int64_t)b * H + ph) * W + pw) * ldy
                                                                                  : g_zero_pixel);      // unconditional load
                        const float* wt = filt + (int64_t)((dh * K + dw) * CT + C0) * ldf;
#pragma unroll
                        for (int c = 0; c < C1; ++c) acc[c] += g * wt[c * ldf];
                    }
        float* o = d1 + e * ld1;
#pragma unroll
        for (int c = 0; c < C1; c += 4) {
            float4 v = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
            if (accumulate) {
                const float4 p = *reinterpret_cast<const float4*>(o + c);
                v.x += p.x, v.y += p.y, v.z += p.z, v.w += p.w;
            }
            *reinterpret_cast<float4*>(o + c) = v;
        }
    }
}

// After the input-gradient convolution (dX of concat(src0, up2x(src1)) as one [R][C0 + C1] matrix):
// channels [0, C0) go to the full-resolution source, channels [C0, C0 + C1) are summed over each 2x2
// block into the coarse source (the adjoint of nearest-neighbour up-sampling); "=" or "+=" per target.
__global__ __launch_bounds__(TPB) void split_sumpool_kernel(const float* __restrict__ dx, int ldx, float* __restrict__ d0,
                                                            int C0, int ld0, int acc0, float* __restrict__ d1, int C1,
                                                            int ld1, int acc1, int B, int H, int W) {
    const int q0 = d0 ? ld0 >> 2 : 0, q1 = d1 ? ld1 >> 2 : 0;
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t n0 = (int64_t)B * H * W * q0, n1 = (int64_t)B * H2 * W2 * q1;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n0 + n1; e += (int64_t)gridDim.x * TPB) {
        if (e < n0) {
            const int c = (int)(e % q0) * 4;
            const int64_t px = e / q0;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C0) v = *reinterpret_cast<const float4*>(dx + px * ldx + c);
            float4* o = reinterpret_cast<float4*>(d0 + px * ld0 + c);
            if (acc0) {
                const float4 p = *o;
                v.x += p.x, v.y += p.y, v.z += p.z, v.w += p.w;
            }
            *o = v;
        } else {
            const int64_t e1 = e - n0;
            const int c = (int)(e1 % q1) * 4;
            int64_t px = e1 / q1;
            const int w2 = (int)(px % W2);
            px /= W2;
            const int h2 = (int)(px % H2), b = (int)(px / H2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C1) {
#pragma unroll
                for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                    for (int dw = 0; dw < 2; ++dw) {
                        const float4 t = *reinterpret_cast<const float4*>(
                            dx + (((int64_t)b * H + 2 * h2 + dh) * W + 2 * w2 + dw) * ldx + C0 + c);
                        v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
                    }
            }
            float4* o = reinterpret_cast<float4*>(d1 + (((int64_t)b * H2 + h2) * W2 + w2) * ld1 + c);
            if (acc1) {
                const float4 p = *o;
                v.x += p.x, v.y += p.y, v.z += p.z, v.w += p.w;
            }
            *o = v;
        }
    }
}

// which = 0: gradient of the full-resolution source (channels [0, C0)); which = 1: of the up-sampled one
__global__ __launch_bounds__(TPB) void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dst,
                                                     const ConvGeom g, const int which, const int accumulate) {
    const int Ct = g.C0 + g.C1, p = g.k / 2;
    const int Hd = which ? g.H >> 1 : g.H, Wd = which ? g.W >> 1 : g.W;
    const int Cd = which ? g.C1 : g.C0, ldd = which ? g.ld1 : g.ld0, coff = which ? g.C0 : 0;
    const int64_t n = (int64_t)g.B * Hd * Wd * ldd;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int c = (int)(e % ldd);
        int64_t pix = e / ldd;
        const int wd = (int)(pix % Wd);
        pix /= Wd;
        const int hd = (int)(pix % Hd), b = (int)(pix / Hd);
        float acc = 0.f;
        if (c < Cd) {
            const int reps = which ? 2 : 1;
            for (int dh = 0; dh < reps; ++dh)
                for (int dw = 0; dw < reps; ++dw) {
                    const int h = which ? 2 * hd + dh : hd, w = which ? 2 * wd + dw : wd;
                    for (int tap = 0; tap < g.k * g.k; ++tap) {
                        // output pixel (ho, wo) read this input pixel through tap (kh, kw): ho + kh - p = h
                        const int ho = h - tap / g.k + p, wo = w - tap % g.k + p;
                        if (ho >= 0 && ho < g.H && wo >= 0 && wo < g.W)
                            acc += dcol[(((int64_t)b * g.H + ho) * g.W + wo) * g.Kc + tap * Ct + coff + c];
                    }
                }
        }
        dst[e] = accumulate ? dst[e] + acc : acc;
    }
}

// two-stage column reduction of a pair of per-element quantities
//   MODE 0: (x, x^2)            -> batch-norm statistics
//   MODE 1: (g, g * xhat), g = dy * act'(gamma xhat + beta)  -> d beta, d gamma
struct BnArgs {
    const float* x;
    const float* dy;
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
    int64_t R;
    int C, ld, act, has_bn;
};

__device__ __forceinline__ float act_grad(float z, int act) {
    if (act == 1) return z > 0.f ? 1.f : 0.f;
    if (act == 2) return z > 0.f ? 1.f : 0.2f;
    if (act == 3) return z > 0.f ? 1.f : 0.3f;
    return 1.f;
}

// Rows are read whole with 16-byte loads: thread t owns the float4 column group t % (ld / 4) of rows
// t / (ld / 4), + TPB / (ld / 4), ...; the row-threads of a column group are then summed through LDS.
// (Tried and dropped: letting the last block to arrive run the final stage -- one launch per layer instead of two.
// The device-scope fences every block then needs cost more than the second launch: 72 us against 36 + 39.)
template <int MODE>
__global__ __launch_bounds__(TPB) void colpair_partial_kernel(const BnArgs a, float* __restrict__ part) {
    __shared__ float red[2][TPB * 4];
    const int q4 = a.ld >> 2;                              // float4 groups per row (<= 64: ld <= 256)
    const int rp = TPB / q4;                               // rows per pass
    const int cq = threadIdx.x % q4, rl = threadIdx.x / q4;
    const int64_t chunk = (a.R + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = min(a.R, r0 + chunk);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f}, ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1 && a.has_bn)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * cq + k;
            if (c < a.C) mu[k] = a.mean[c], rs[k] = a.rstd[c], ga[k] = a.gamma[c], be[k] = a.beta[c];
        }
    if (rl < rp) {
        int64_t r = r0 + rl;
        if (MODE == 0) {
            // statistics: four rows per trip, their loads issued together (the sums stay in row order)
            for (; r + 3 * (int64_t)rp < r1; r += 4 * (int64_t)rp) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(a.x + (r + u * (int64_t)rp) * a.ld + 4 * cq);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float xv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) s1[k] += xv[k], s2[k] += xv[k] * xv[k];
                }
            }
        }
        for (; r < r1; r += rp) {
            const float4 xv4 = *reinterpret_cast<const float4*>(a.x + r * a.ld + 4 * cq);
            const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) s1[k] += xv[k], s2[k] += xv[k] * xv[k];
            } else {
                const float4 dy4 = *reinterpret_cast<const float4*>(a.dy + r * a.ld + 4 * cq);
                const float dy[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (xv[k] - mu[k]) * rs[k];
                    const float g = dy[k] * act_grad(a.has_bn ? ga[k] * xh + be[k] : xv[k], a.act);
                    s1[k] += g, s2[k] += g * xh;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[0][threadIdx.x * 4 + k] = s1[k], red[1][threadIdx.x * 4 + k] = s2[k];
    __syncthreads();
    // thread c < C sums the rp row-threads of its channel (fixed order: deterministic)
    const int c = threadIdx.x;
    if (c < a.C) {
        float t1 = 0.f, t2 = 0.f;
        for (int j = 0; j < rp; ++j) {
            const int src = (j * q4 + (c >> 2)) * 4 + (c & 3);
            t1 += red[0][src], t2 += red[1][src];
        }
        part[((int64_t)blockIdx.y * 2 + 0) * a.C + c] = t1;
        part[((int64_t)blockIdx.y * 2 + 1) * a.C + c] = t2;
    }
}

// MODE 0: out0 = mean, out1 = rstd.  MODE 1: out0 = sum g (d beta), out1 = sum g xhat (d gamma).
// One wave per channel: lane l adds partials l, l + 64, ... in double, then a fixed shuffle tree
// (deterministic; the serial one-thread-per-channel loop over 256 partials took 39 us per launch).
template <int MODE>
__global__ __launch_bounds__(64) void colpair_final_kernel(const float* __restrict__ part, int parts, int C, int64_t R,
                                                           float eps, float* __restrict__ out0,
                                                           float* __restrict__ out1) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    // eight partials per lane and trip, their sixteen loads in flight together (one by one a lane's strided loads were
    // sixteen memory latencies in a row at 512 partials: 16 us per launch, eleven launches per U-Net step); added in index order
    for (int p0 = lane; p0 < parts; p0 += 512) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 64 * u;
            const int pc = p < parts ? p : p0;
            a[u] = part[((int64_t)pc * 2 + 0) * C + c];
            b[u] = part[((int64_t)pc * 2 + 1) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p0 + 64 * u < parts) s1 += (double)a[u], s2 += (double)b[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (lane != 0) return;
    if (MODE == 0) {
        const double m = s1 / (double)R;
        double var = s2 / (double)R - m * m;
        if (var < 0.0) var = 0.0;
        out0[c] = (float)m;
        out1[c] = (float)(1.0 / sqrt(var + (double)eps));
    } else {
        out0[c] = (float)s1;
        out1[c] = (float)s2;
    }
}

__global__ __launch_bounds__(TPB) void bn_act_kernel(const BnArgs a, float* __restrict__ y) {
    const int64_t n = a.R * a.ld;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int c = (int)(e % a.ld);
        float v = 0.f;
        if (c < a.C) {
            v = a.x[e];
            if (a.has_bn) v = a.gamma[c] * ((v - a.mean[c]) * a.rstd[c]) + a.beta[c];
            if (a.act == 1) v = fmaxf(v, 0.f);
            if (a.act == 2) v = v > 0.f ? v : 0.2f * v;
            if (a.act == 3) v = v > 0.f ? v : 0.3f * v;
        }
        y[e] = v;
    }
}

__device__ __forceinline__ float bn_act_one(float v, float sc, float sh, int act) {
    v = fmaf(v, sc, sh);
    if (act == 1) v = fmaxf(v, 0.f);
    if (act == 2) v = v > 0.f ? v : 0.2f * v;
    if (act == 3) v = v > 0.f ? v : 0.3f * v;
    return v;
}

// 16-byte form of bn_act_kernel (ld % 4 == 0): thread <-> four channels of one pixel; batch norm folded into one
// multiply-add per element (scale = gamma rstd, shift = beta - mean scale; padding channels: scale = shift = 0).
// POOL: thread <-> four channels of one POOLED pixel: the four pixels of its 2 x 2 window are normalised and
// activated, written to y (when y is given: training keeps the activation for the backward pass) and their
// maximum to `pooled` -- the separate pooling pass (one more read of the activation) is gone.
template <bool POOL>
__global__ __launch_bounds__(TPB) void bn_act4_kernel(const BnArgs a, float* __restrict__ y, float* __restrict__ pooled, int B,
                                                      int H, int W) {
    const int q4 = a.ld >> 2;
    const int64_t pixels = POOL ? (int64_t)B * (H >> 1) * (W >> 1) : a.R;
    const int64_t n = pixels * q4;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int cq = (int)(e % q4);
        const int64_t pix = e / q4;
        float sc[4], sh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * cq + k;
            sc[k] = c < a.C ? 1.f : 0.f, sh[k] = 0.f;
            if (a.has_bn && c < a.C) {
                sc[k] = a.gamma[c] * a.rstd[c];
                sh[k] = a.beta[c] - a.mean[c] * sc[k];
            }
        }
        if (!POOL) {
            const float4 v = *reinterpret_cast<const float4*>(a.x + pix * a.ld + 4 * cq);
            *reinterpret_cast<float4*>(y + pix * a.ld + 4 * cq) = make_float4(
                bn_act_one(v.x, sc[0], sh[0], a.act), bn_act_one(v.y, sc[1], sh[1], a.act), bn_act_one(v.z, sc[2], sh[2], a.act),
                bn_act_one(v.w, sc[3], sh[3], a.act));
        } else {
            const int W2 = W >> 1, H2 = H >> 1;
            const int w2 = (int)(pix % W2);
            const int64_t bh = pix / W2;
            const int h2 = (int)(bh % H2);
            const int64_t b = bh / H2;
            const int64_t o00 = ((b * H + 2 * h2) * W + 2 * w2) * a.ld + 4 * cq;
            const int64_t offs[4] = {0, a.ld, (int64_t)W * a.ld, (int64_t)W * a.ld + a.ld};
            float4 in[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) in[q] = *reinterpret_cast<const float4*>(a.x + o00 + offs[q]);
            float4 best;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 r = make_float4(bn_act_one(in[q].x, sc[0], sh[0], a.act), bn_act_one(in[q].y, sc[1], sh[1], a.act),
                                             bn_act_one(in[q].z, sc[2], sh[2], a.act), bn_act_one(in[q].w, sc[3], sh[3], a.act));
                if (y) *reinterpret_cast<float4*>(y + o00 + offs[q]) = r;
                best = q == 0 ? r : make_float4(fmaxf(best.x, r.x), fmaxf(best.y, r.y), fmaxf(best.z, r.z), fmaxf(best.w, r.w));
            }
            *reinterpret_cast<float4*>(pooled + pix * a.ld + 4 * cq) = best;
        }
    }
}

__global__ __launch_bounds__(TPB) void bn_act_bwd_apply_kernel(const BnArgs a, const float* __restrict__ sum_g,
                                                               const float* __restrict__ sum_gx,
                                                               float* __restrict__ dx) {
    const int64_t n = a.R * a.ld;
    const float invn = 1.f / (float)a.R;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int c = (int)(e % a.ld);
        float v = 0.f;
        if (c < a.C) {
            const float xv = a.x[e];
            if (a.has_bn) {
                const float rs = a.rstd[c], ga = a.gamma[c];
                const float xh = (xv - a.mean[c]) * rs;
                const float g = a.dy[e] * act_grad(ga * xh + a.beta[c], a.act);
                v = ga * rs * (g - sum_g[c] * invn - xh * sum_gx[c] * invn);
            } else {
                v = a.dy[e] * act_grad(xv, a.act);
            }
        }
        dx[e] = v;  // padding columns are written as zeros: dx is a GEMM operand next
    }
}

// Backward of an ENCODER layer in one piece: act(bn(conv)) was max-pooled 2 x 2 (bn_act4_kernel<true>), and the gradient
// arrives for the POOLED output.  The full-resolution activation and its gradient are never materialised: a thread takes
// four channels of one pooled pixel, recomputes the four activations of its window from `conv` exactly as the forward
// pass did (bn_act_one: the same bits, so the same first-maximum-wins choice as maxpool2_bwd_kernel), and routes the
// pooled gradient to that pixel.
//   APPLY = false: partial sums (sum g, sum g xhat) over the thread's pixels -> part [parts][2][C]   (d beta, d gamma)
//   APPLY = true : dx = gamma rstd (g - sum_g / N - xhat sum_gx / N) for the four pixels (without batch norm: dx = g), and
//                  the partial column sums of dx -> part [parts][2][C] (row 0; the bias gradient of a layer without batch norm)
// Replaces maxpool2_bwd + colpair_partial<1> + bn_act_bwd_apply + the column sum, which between them read the activation
// once, wrote and read its gradient three times and read dx once more: 1.0 of 19 ms per training step at 512 clips.
template <bool APPLY>
__global__ __launch_bounds__(TPB) void bn_act_pool_bwd_kernel(const BnArgs a, int B, int H, int W,
                                                              const float* __restrict__ sum_g,
                                                              const float* __restrict__ sum_gx, float* __restrict__ dx,
                                                              float* __restrict__ part) {
    __shared__ float red[2][TPB * 4];
    const int q4 = a.ld >> 2;
    const int rp = TPB / q4;
    const int cq = threadIdx.x % q4, rl = threadIdx.x / q4;
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t P = (int64_t)B * H2 * W2;                 // pooled pixels
    const int64_t chunk = (P + gridDim.y - 1) / gridDim.y;
    const int64_t p0 = (int64_t)blockIdx.y * chunk, p1 = min(P, p0 + chunk);
    const float invn = 1.f / (float)a.R;
    float sc[4], sh[4], mu[4], rs[4], ga[4], be[4], mg[4], mgx[4];
    bool live[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * cq + k;
        live[k] = c < a.C;
        sc[k] = live[k] ? 1.f : 0.f, sh[k] = 0.f, mu[k] = 0.f, rs[k] = 1.f, ga[k] = 1.f, be[k] = 0.f, mg[k] = 0.f, mgx[k] = 0.f;
        if (a.has_bn && live[k]) {
            mu[k] = a.mean[c], rs[k] = a.rstd[c], ga[k] = a.gamma[c], be[k] = a.beta[c];
            sc[k] = ga[k] * rs[k];
            sh[k] = be[k] - mu[k] * sc[k];
            if (APPLY) mg[k] = sum_g[c] * invn, mgx[k] = sum_gx[c] * invn;
        }
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < rp) {
#pragma unroll 2
        for (int64_t pix = p0 + rl; pix < p1; pix += rp) {
            const int w2 = (int)(pix % W2);
            const int64_t bh = pix / W2;
            const int h2 = (int)(bh % H2);
            const int64_t b = bh / H2;
            const int64_t o00 = ((b * H + 2 * h2) * W + 2 * w2) * a.ld + 4 * cq;
            const int64_t offs[4] = {0, a.ld, (int64_t)W * a.ld, (int64_t)W * a.ld + a.ld};
            float4 in4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) in4[q] = *reinterpret_cast<const float4*>(a.x + o00 + offs[q]);
            const float4 d4 = *reinterpret_cast<const float4*>(a.dy + pix * a.ld + 4 * cq);
            const float d[4] = {d4.x, d4.y, d4.z, d4.w};
            float out[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xv[4] = {(&in4[0].x)[k], (&in4[1].x)[k], (&in4[2].x)[k], (&in4[3].x)[k]};
                int best = 0;
                float bv = bn_act_one(xv[0], sc[k], sh[k], a.act);
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const float y = bn_act_one(xv[q], sc[k], sh[k], a.act);
                    if (y > bv) bv = y, best = q;           // first maximum in row-major window order wins ties
                }
                const float xb = best == 0 ? xv[0] : (best == 1 ? xv[1] : (best == 2 ? xv[2] : xv[3]));
                const float xhb = (xb - mu[k]) * rs[k];
                const float g = live[k] ? d[k] * act_grad(a.has_bn ? ga[k] * xhb + be[k] : xb, a.act) : 0.f;
                if (!APPLY) {
                    s1[k] += g, s2[k] += g * xhb;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float gq = q == best ? g : 0.f;
                        float v = gq;
                        if (a.has_bn) v = live[k] ? sc[k] * (gq - mg[k] - (xv[q] - mu[k]) * rs[k] * mgx[k]) : 0.f;
                        out[q][k] = v;
                        s1[k] += v;
                    }
                }
            }
            if (APPLY) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(dx + o00 + offs[q]) = make_float4(out[q][0], out[q][1], out[q][2], out[q][3]);
            }
        }
    }
    if (!part) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) red[0][threadIdx.x * 4 + k] = s1[k], red[1][threadIdx.x * 4 + k] = s2[k];
    __syncthreads();
    const int c = threadIdx.x;
    if (c < a.C) {
        float t1 = 0.f, t2 = 0.f;
        for (int j = 0; j < rp; ++j) {
            const int src = (j * q4 + (c >> 2)) * 4 + (c & 3);
            t1 += red[0][src], t2 += red[1][src];
        }
        part[((int64_t)blockIdx.y * 2 + 0) * a.C + c] = t1;
        part[((int64_t)blockIdx.y * 2 + 1) * a.C + c] = t2;
    }
}

__global__ __launch_bounds__(TPB) void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                       int W, int C, int ld) {
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t n = (int64_t)B * H2 * W2 * ld;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int c = (int)(e % ld);
        int64_t pix = e / ld;
        const int w2 = (int)(pix % W2);
        pix /= W2;
        const int h2 = (int)(pix % H2), b = (int)(pix / H2);
        float v = 0.f;
        if (c < C) {
            const float* p = x + (((int64_t)b * H + 2 * h2) * W + 2 * w2) * ld + c;
            v = fmaxf(fmaxf(p[0], p[ld]), fmaxf(p[(int64_t)W * ld], p[(int64_t)W * ld + ld]));
        }
        y[e] = v;
    }
}

__global__ __launch_bounds__(TPB) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int B, int H, int W, int C, int ld) {
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t n = (int64_t)B * H2 * W2 * ld;
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int c = (int)(e % ld);
        int64_t pix = e / ld;
        const int w2 = (int)(pix % W2);
        pix /= W2;
        const int h2 = (int)(pix % H2), b = (int)(pix / H2);
        const int64_t o = (((int64_t)b * H + 2 * h2) * W + 2 * w2) * ld + c;
        const int64_t offs[4] = {0, ld, (int64_t)W * ld, (int64_t)W * ld + ld};
        int best = 0;
        if (c < C) {
            float bv = x[o];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const float v = x[o + offs[q]];
                if (v > bv) bv = v, best = q;  // first maximum in row-major window order wins ties
            }
        }
        const float g = c < C ? dy[e] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) dx[o + offs[q]] = (q == best) ? g : 0.f;
    }
}

// Row slabs of the column reductions: one workgroup each.  Four per CU for the large activations, four rows in
// flight per thread (with 256 slabs and one row per pass a block streamed 512 KB through 256 threads' 16-byte loads,
// one memory latency per pass: 1.3 TB/s on the 134 MB activation of the widest decoder layer).
constexpr int MAXPARTS = 1024;
inline int parts_for(int64_t R) {
    int64_t p = avsi_ceil_div(R, 128);      // (512 until round 5: the deep layers' 8192 rows were 16 workgroups)
    return (int)(p < 1 ? 1 : (p > MAXPARTS ? MAXPARTS : p));
}
}  // namespace

static int check_geom(int B, int H, int W, int C0, int ld0, int C1, int ld1, int k, int Kc) {
    if (B <= 0 || H <= 0 || W <= 0 || C0 < 0 || C1 < 0 || C0 + C1 <= 0 || k < 1 || !(k & 1)) return AVSI_ERR_INVALID_ARG;
    if (Kc < k * k * (C0 + C1) || (C0 && ld0 < C0) || (C1 && ld1 < C1)) return AVSI_ERR_INVALID_ARG;
    if (C1 && ((H | W) & 1)) return AVSI_ERR_INVALID_ARG;
    return AVSI_OK;
}

extern "C" int avsi_im2col_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H,
                               int W, int k, float* col, int Kc, void* stream) {
    const int rc = check_geom(B, H, W, C0, ld0, C1, ld1, k, Kc);
    if (rc != AVSI_OK || !col || (C0 && !src0) || (C1 && !src1_coarse)) return rc != AVSI_OK ? rc : AVSI_ERR_INVALID_ARG;
    const ConvGeom g{B, H, W, C0, ld0, C1, ld1, k, Kc};
    avsi_clear_error();
    const bool vec4 = !(C0 & 3) && !(C1 & 3) && !(ld0 & 3) && !(ld1 & 3) && !(Kc & 3) &&
                      !((reinterpret_cast<uintptr_t>(src0) | reinterpret_cast<uintptr_t>(src1_coarse) |
                         reinterpret_cast<uintptr_t>(col)) & 15);
    if (vec4)
        hipLaunchKernelGGL(im2col4_kernel, dim3(grid_for((int64_t)B * H * W * (Kc / 4))), dim3(TPB), 0, (hipStream_t)stream,
                           src0, src1_coarse, col, g);
    else
        hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((int64_t)B * H * W * Kc)), dim3(TPB), 0, (hipStream_t)stream, src0,
                           src1_coarse, col, g);
    return avsi_launch_status();
}

extern "C" int avsi_conv2d_thin_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H,
                                    int W, int k, const float* filter, int ldf, const float* bias, int Cout, float* out,
                                    int ldo, void* stream) {
    if (!filter || !out || B <= 0 || H <= 0 || W <= 0 || (C0 && !src0) || (C1 && !src1_coarse) || ldf < Cout || ldo < Cout ||
        (C0 && ld0 < C0) || (C1 && ld1 < C1))
        return AVSI_ERR_INVALID_ARG;
    if ((C1 && (((H | W) & 1) || (ld1 & 3) || (reinterpret_cast<uintptr_t>(src1_coarse) & 15))) ||
        ((Cout & 3) == 0 && ((ldo & 3) || (reinterpret_cast<uintptr_t>(out) & 15))))
        return AVSI_ERR_UNSUPPORTED;
    const dim3 grid(grid_for((int64_t)B * H * W)), block(TPB);
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    if (k == 7 && C0 == 1 && C1 == 0 && Cout == 16 && H % 16 == 0 && W % 64 == 0)
        hipLaunchKernelGGL((conv7_c1_mfma_kernel<false>), dim3(B * (H / 16) * (W / 64)), dim3(256), 0, st, src0, ld0, filter, ldf,
                           bias, out, ldo, H, W);
    else if (k == 7 && C0 == 1 && C1 == 0 && Cout == 16)
        hipLaunchKernelGGL((direct_conv_kernel<7, 1, 0, 16>), grid, block, 0, st, src0, ld0, src1_coarse, ld1, filter, ldf, bias,
                           out, ldo, B, H, W);
    else if (k == 3 && C0 == 1 && C1 == 16 && Cout == 1 && H % 8 == 0 && W % 32 == 0)
        hipLaunchKernelGGL((conv3_c17_out1_kernel<false, false>), dim3(B * (H / 8) * (W / 32)), dim3(256), 0, st, src0, ld0, src1_coarse,
                           ld1, filter, ldf, bias, out, ldo, H, W, (float*)nullptr, Bn4{});
    else if (k == 3 && C0 == 1 && C1 == 16 && Cout == 1)
        hipLaunchKernelGGL((direct_conv_kernel<3, 1, 16, 1>), grid, block, 0, st, src0, ld0, src1_coarse, ld1, filter, ldf, bias,
                           out, ldo, B, H, W);
    else if (k == 1 && C0 == 1 && C1 == 0 && Cout == 1)
        hipLaunchKernelGGL((direct_conv_kernel<1, 1, 0, 1>), grid, block, 0, st, src0, ld0, src1_coarse, ld1, filter, ldf, bias,
                           out, ldo, B, H, W);
    else
        return AVSI_ERR_UNSUPPORTED;
    return avsi_launch_status();
}

// Inference tail of the U-Net (models.py:605-615 in one call, three launches): d6 = conv3x3(concat(src0, up2x(src1)), 17 -> 1)
// with its batch statistics taken in the convolution's own epilogue, then ONE pass: batch norm + LeakyReLU(0.2) + the 1 x 1
// output convolution (scalar w_out, b_out) + the sequence mask.  `conv` [B*H*W] (pitch 1) is scratch; `logits` may be null.
extern "C" size_t avsi_unet_tail_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || H % 8 || W % 32) return 0;
    return (size_t)B * (H / 8) * (W / 32) * 2 * sizeof(float) + 2 * sizeof(float);
}

extern "C" int avsi_unet_tail_f32(const float* src0, int ld0, const float* src1_coarse, int ld1, const float* const* src1_bn, int B,
                                  int H, int W, const float* filter, int ldf, const float* bias, const float* gamma,
                                  const float* beta, float eps, const float* w_out, const float* b_out, const long long* seq_len,
                                  float* conv, float* logits, float* pred, void* workspace, size_t workspace_bytes, void* stream) {
    if (!src0 || !src1_coarse || !filter || !gamma || !beta || !w_out || !seq_len || !conv || !pred || B <= 0 || ld0 < 1 || ld1 < 16 ||
        ldf < 1)
        return AVSI_ERR_INVALID_ARG;
    if (H % 8 || W % 32 || (ld1 & 3) || ((reinterpret_cast<uintptr_t>(src1_coarse) | reinterpret_cast<uintptr_t>(conv) |
                                           reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(logits)) & 15))
        return AVSI_ERR_UNSUPPORTED;
    const size_t need = avsi_unet_tail_workspace_bytes(B, H, W);
    if (!workspace || workspace_bytes < need) return AVSI_ERR_WORKSPACE;
    const int blocks = B * (H / 8) * (W / 32);
    float* part = static_cast<float*>(workspace);
    float* mean = part + 2 * (size_t)blocks;
    const hipStream_t st = (hipStream_t)stream;
    const int64_t R = (int64_t)B * H * W;
    avsi_clear_error();
    if (src1_bn) {
        if (!src1_bn[0] || !src1_bn[1] || !src1_bn[2] || !src1_bn[3]) return AVSI_ERR_INVALID_ARG;
        hipLaunchKernelGGL((conv3_c17_out1_kernel<true, true>), dim3(blocks), dim3(256), 0, st, src0, ld0, src1_coarse, ld1, filter, ldf,
                           bias, conv, 1, H, W, part, Bn4{src1_bn[0], src1_bn[1], src1_bn[2], src1_bn[3]});
    } else {
        hipLaunchKernelGGL((conv3_c17_out1_kernel<true, false>), dim3(blocks), dim3(256), 0, st, src0, ld0, src1_coarse, ld1, filter, ldf,
                           bias, conv, 1, H, W, part, Bn4{});
    }
    hipLaunchKernelGGL(colpair_final_kernel<0>, dim3(1), dim3(64), 0, st, (const float*)part, blocks, 1, R, eps, mean, mean + 1);
    hipLaunchKernelGGL(unet_tail_kernel, dim3(grid_for(R / 4)), dim3(TPB), 0, st, (const float*)conv, (const float*)mean,
                       (const float*)(mean + 1), gamma, beta, w_out, b_out, seq_len, logits, pred, R / 4, H, W);
    return avsi_launch_status();
}

int avsi_sum_slabs_launch(const float* slabs, int64_t n, int count, int64_t stride, float* out, float alpha, hipStream_t st);

static int thin_wgrad_blocks(int64_t pixels) {
    const int64_t want = avsi_ceil_div(pixels, (int64_t)TPB * 8);
    return (int)(want < 1 ? 1 : (want > 512 ? 512 : want));
}

extern "C" size_t avsi_conv2d_thin_wgrad_workspace_bytes(int C0, int C1, int k, int Cout, int B, int H, int W) {
    return (size_t)thin_wgrad_blocks((int64_t)B * H * W) * (size_t)(k * k * (C0 + C1) * Cout) * sizeof(float);
}

// dw [k*k*(C0+C1)][ldw] for the thin layers (same (k, C0, C1, Cout) set as avsi_conv2d_thin_f32); Cout == ldw or Cout == 1
extern "C" int avsi_conv2d_thin_wgrad_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B,
                                          int H, int W, int k, const float* dy, int ldy, int Cout, float* dw, int ldw,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !dw || B <= 0 || H <= 0 || W <= 0 || (C0 && !src0) || (C1 && !src1_coarse) || ldy < Cout || ldw < Cout)
        return AVSI_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < avsi_conv2d_thin_wgrad_workspace_bytes(C0, C1, k, Cout, B, H, W)) return AVSI_ERR_WORKSPACE;
    if (C1 && (((H | W) & 1) || (ld1 & 3) || (reinterpret_cast<uintptr_t>(src1_coarse) & 15))) return AVSI_ERR_UNSUPPORTED;
    const int blocks = thin_wgrad_blocks((int64_t)B * H * W);
    const int na = k * k * (C0 + C1) * Cout;
    const hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    avsi_clear_error();
    static const bool c7_mfma = !(getenv("AVSI_CONV7_WGRAD_MFMA") && atoi(getenv("AVSI_CONV7_WGRAD_MFMA")) == 0);
    if (k == 7 && C0 == 1 && C1 == 0 && Cout == 16 && c7_mfma && H % 8 == 0 && W % 32 == 0 && !(ldy & 3) &&
        !(reinterpret_cast<uintptr_t>(dy) & 15) && B * (H / 8) * (W / 32) >= blocks)
        // (one partial filter per workgroup, `blocks` of them -- the workspace and the sum below are sized for that)
        hipLaunchKernelGGL(conv7_c1_wgrad_mfma_kernel, dim3(blocks), dim3(256), 0, st, src0, ld0, dy, ldy, part, H, W,
                           B * (H / 8) * (W / 32));
    else if (k == 7 && C0 == 1 && C1 == 0 && Cout == 16)
        hipLaunchKernelGGL((thin_wgrad_kernel<7, 1, 0, 16>), dim3(blocks, 7), dim3(TPB), 0, st, src0, ld0, src1_coarse, ld1, dy, ldy,
                           part, B, H, W);
    else if (k == 3 && C0 == 1 && C1 == 16 && Cout == 1 && c7_mfma && H % 8 == 0 && W % 32 == 0 && !(ld1 & 3) &&
             !(reinterpret_cast<uintptr_t>(src1_coarse) & 15) && B * (H / 8) * (W / 32) >= blocks)
        hipLaunchKernelGGL(conv3_c17_wgrad_mfma_kernel, dim3(blocks), dim3(256), 0, st, src0, ld0, src1_coarse, ld1, dy, ldy, part, H, W,
                           B * (H / 8) * (W / 32));
    else if (k == 3 && C0 == 1 && C1 == 16 && Cout == 1)
        hipLaunchKernelGGL((thin_wgrad_kernel<3, 1, 16, 1>), dim3(blocks, 3), dim3(TPB), 0, st, src0, ld0, src1_coarse, ld1, dy, ldy,
                           part, B, H, W);
    else if (k == 1 && C0 == 1 && C1 == 0 && Cout == 1)
        hipLaunchKernelGGL((thin_wgrad_kernel<1, 1, 0, 1>), dim3(blocks, 1), dim3(TPB), 0, st, src0, ld0, src1_coarse, ld1, dy, ldy,
                           part, B, H, W);
    else
        return AVSI_ERR_UNSUPPORTED;
    int rc = avsi_launch_status();
    if (rc != AVSI_OK) return rc;
    if (ldw == Cout) return avsi_sum_slabs_launch(part, na, blocks, na, dw, 1.f, st);
    // Cout = 1 stored with pitch ldw: sum into a dense vector behind the partials, then scatter
    if (Cout != 1) return AVSI_ERR_UNSUPPORTED;
    rc = avsi_sum_slabs_launch(part, na, blocks, na, part, 1.f, st);      // in place: row 0 receives the sum of all rows
    if (rc != AVSI_OK) return rc;
    if (hipMemsetAsync(dw, 0, (size_t)na * ldw * sizeof(float), st) != hipSuccess) return AVSI_ERR_LAUNCH;
    if (hipMemcpy2DAsync(dw, (size_t)ldw * sizeof(float), part, sizeof(float), sizeof(float), na, hipMemcpyDeviceToDevice, st) !=
        hipSuccess)
        return AVSI_ERR_LAUNCH;
    return AVSI_OK;
}

// Filter gradient on the 16-wide MFMA (thin_mfma_wgrad_kernel): the U-Net's three few-channel layers with batch norm,
// (k, C0, C1, Cout) = (3, 16, 32, 16), (5, 16, 0, 32), (3, 32, 64, 32); H % 4 == 0, W % 32 == 0.
extern "C" int avsi_conv2d_thin_mfma_wgrad_supported(int k, int C0, int C1, int Cout, int H, int W) {
    return (H % 4 == 0 && W % 32 == 0) && ((k == 3 && C0 == 16 && C1 == 32 && Cout == 16) ||
                                           (k == 5 && C0 == 16 && C1 == 0 && Cout == 32) ||
                                           (k == 3 && C0 == 32 && C1 == 64 && Cout == 32));
}

static int thin_mfma_wgrad_blocks(int k, int C0, int C1, int n_tiles) {
    // LDS per workgroup: 23 / 35 KB (two per CU), 96 KB for the 96-channel layer (one per CU)
    const int per_cu = (C0 + C1 > 48) ? 1 : 2;
    return n_tiles < per_cu * AVSI_NUM_CU ? n_tiles : per_cu * AVSI_NUM_CU;
}

extern "C" size_t avsi_conv2d_thin_mfma_wgrad_workspace_bytes(int C0, int C1, int k, int Cout, int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)thin_mfma_wgrad_blocks(k, C0, C1, B * (H / 4) * (W / 32)) * (size_t)(k * k * (C0 + C1) * Cout) * sizeof(float);
}

extern "C" int avsi_conv2d_thin_mfma_wgrad_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B,
                                               int H, int W, int k, const float* dy, int ldy, int Cout, float* dw, int ldw,
                                               const float* zeros64, void* workspace, size_t workspace_bytes, void* stream) {
    if (!src0 || !dy || !dw || !zeros64 || B <= 0 || H <= 0 || W <= 0 || (C1 && !src1_coarse) || ldy < Cout || ld0 < C0 ||
        (C1 && ld1 < C1))
        return AVSI_ERR_INVALID_ARG;
    if (!avsi_conv2d_thin_mfma_wgrad_supported(k, C0, C1, Cout, H, W) || ldw != Cout || (ld0 & 3) || (ld1 & 3) || (ldy & 3) ||
        ((reinterpret_cast<uintptr_t>(src0) | reinterpret_cast<uintptr_t>(src1_coarse) | reinterpret_cast<uintptr_t>(dy) |
          reinterpret_cast<uintptr_t>(zeros64) | reinterpret_cast<uintptr_t>(dw)) & 15))
        return AVSI_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < avsi_conv2d_thin_mfma_wgrad_workspace_bytes(C0, C1, k, Cout, B, H, W)) return AVSI_ERR_WORKSPACE;
    const int n_tiles = B * (H / 4) * (W / 32);
    const int blocks = thin_mfma_wgrad_blocks(k, C0, C1, n_tiles);
    const int na = k * k * (C0 + C1) * Cout;
    const hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    avsi_clear_error();
#define AVSI_THIN_WGRAD(KS, CA, CB, CO)                                                                                        \
    do {                                                                                                                      \
        constexpr size_t lds = ((size_t)(4 + KS - 1) * (32 + KS - 1) * (CA + CB + 1) + (size_t)128 * (CO + 1)) * 4;           \
        (void)hipFuncSetAttribute((const void*)thin_mfma_wgrad_kernel<KS, CA, CB, CO>,                                         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
        hipLaunchKernelGGL((thin_mfma_wgrad_kernel<KS, CA, CB, CO>), dim3(blocks), dim3(256), lds, st, src0, ld0, src1_coarse, \
                           ld1, dy, ldy, part, H, W, zeros64, n_tiles);                                                       \
    } while (0)
    if (k == 3 && C0 == 16) AVSI_THIN_WGRAD(3, 16, 32, 16);
    else if (k == 5) AVSI_THIN_WGRAD(5, 16, 0, 32);
    else AVSI_THIN_WGRAD(3, 32, 64, 32);
#undef AVSI_THIN_WGRAD
    const int rc = avsi_launch_status();
    if (rc != AVSI_OK) return rc;
    return avsi_sum_slabs_launch(part, na, blocks, na, dw, 1.f, st);
}

extern "C" int avsi_conv2d_thin_dx_coarse_f32(const float* dy, int ldy, const float* filter, int ldf, float* dsrc1_coarse, int ld1,
                                              int accumulate, int B, int H, int W, void* stream) {
    if (!dy || !filter || !dsrc1_coarse || B <= 0 || H <= 0 || W <= 0 || ldy < 1 || ldf < 1 || ld1 < 16) return AVSI_ERR_INVALID_ARG;
    if (((H | W) & 1) || (ld1 & 3) || (reinterpret_cast<uintptr_t>(dsrc1_coarse) & 15)) return AVSI_ERR_UNSUPPORTED;
    avsi_clear_error();
    hipLaunchKernelGGL(thin_dx_coarse_kernel, dim3(grid_for((int64_t)B * (H / 2) * (W / 2))), dim3(TPB), 0, (hipStream_t)stream, dy,
                       ldy, filter, ldf, dsrc1_coarse, ld1, accumulate, B, H, W);
    return avsi_launch_status();
}

extern "C" int avsi_split_sumpool_f32(const float* dx, int ldx, float* dsrc0, int C0, int ld0, int accumulate0,
                                      float* dsrc1_coarse, int C1, int ld1, int accumulate1, int B, int H, int W,
                                      void* stream) {
    if (!dx || B <= 0 || H <= 0 || W <= 0 || C0 < 0 || C1 < 0 || ldx < C0 + C1 || (!dsrc0 && !dsrc1_coarse))
        return AVSI_ERR_INVALID_ARG;
    if ((dsrc0 && ld0 < C0) || (dsrc1_coarse && (ld1 < C1 || ((H | W) & 1)))) return AVSI_ERR_INVALID_ARG;
    if ((C0 & 3) || (C1 & 3) || (ldx & 3) || (dsrc0 && (ld0 & 3)) || (dsrc1_coarse && (ld1 & 3)) ||
        ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dsrc0) | reinterpret_cast<uintptr_t>(dsrc1_coarse)) & 15))
        return AVSI_ERR_UNSUPPORTED;
    const int64_t n = (dsrc0 ? (int64_t)B * H * W * (ld0 >> 2) : 0) + (dsrc1_coarse ? (int64_t)B * (H >> 1) * (W >> 1) * (ld1 >> 2) : 0);
    avsi_clear_error();
    hipLaunchKernelGGL(split_sumpool_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, dx, ldx, dsrc0, C0, ld0,
                       accumulate0, dsrc1_coarse, C1, ld1, accumulate1, B, H, W);
    return avsi_launch_status();
}

extern "C" int avsi_col2im_f32(const float* dcol, int Kc, float* dsrc0, int C0, int ld0, float* dsrc1_coarse, int C1,
                               int ld1, int B, int H, int W, int k, int accumulate0, int accumulate1, void* stream) {
    const int rc = check_geom(B, H, W, C0, ld0, C1, ld1, k, Kc);
    if (rc != AVSI_OK || !dcol) return rc != AVSI_OK ? rc : AVSI_ERR_INVALID_ARG;
    const ConvGeom g{B, H, W, C0, ld0, C1, ld1, k, Kc};
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    if (dsrc0 && C0)
        hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((int64_t)B * H * W * ld0)), dim3(TPB), 0, st, dcol, dsrc0, g, 0,
                           accumulate0);
    if (dsrc1_coarse && C1)
        hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * ld1)), dim3(TPB), 0, st, dcol,
                           dsrc1_coarse, g, 1, accumulate1);
    return avsi_launch_status();
}

extern "C" size_t avsi_unet_workspace_bytes(int C) { return ((size_t)MAXPARTS * 2 + 1) * (size_t)C * sizeof(float); }

extern "C" int avsi_colstats_f32(const float* x, int64_t R, int C, int ld, float eps, float* mean, float* rstd,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !mean || !rstd || R <= 0 || C <= 0 || ld < C) return AVSI_ERR_INVALID_ARG;
    if ((ld & 3) || ld > 256 || (reinterpret_cast<uintptr_t>(x) & 15)) return AVSI_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < avsi_unet_workspace_bytes(C)) return AVSI_ERR_WORKSPACE;
    BnArgs a{x, nullptr, nullptr, nullptr, nullptr, nullptr, R, C, ld, 0, 0};
    const int parts = parts_for(R);
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    hipLaunchKernelGGL((colpair_partial_kernel<0>), dim3(1, parts), dim3(TPB), 0, st, a,
                       (float*)workspace);
    hipLaunchKernelGGL(colpair_final_kernel<0>, dim3(C), dim3(64), 0, st,
                       (const float*)workspace, parts, C, R, eps, mean, rstd);
    return avsi_launch_status();
}

extern "C" int avsi_bn_act_f32(const float* x, int64_t R, int C, int ld, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int act, float* y, void* stream) {
    if (!x || !y || R <= 0 || C <= 0 || ld < C || act < 0 || act > 3) return AVSI_ERR_INVALID_ARG;
    const int has_bn = mean != nullptr;
    if (has_bn && (!rstd || !gamma || !beta)) return AVSI_ERR_INVALID_ARG;
    BnArgs a{x, nullptr, mean, rstd, gamma, beta, R, C, ld, act, has_bn};
    avsi_clear_error();
    if ((ld & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
        hipLaunchKernelGGL((bn_act4_kernel<false>), dim3(grid_for(R * (ld / 4))), dim3(TPB), 0, (hipStream_t)stream, a, y, nullptr, 0,
                           0, 0);
    else
        hipLaunchKernelGGL(bn_act_kernel, dim3(grid_for(R * ld)), dim3(TPB), 0, (hipStream_t)stream, a, y);
    return avsi_launch_status();
}

extern "C" int avsi_bn_act_pool_f32(const float* x, int B, int H, int W, int C, int ld, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, int act, float* y, float* pooled, void* stream) {
    if (!x || !pooled || B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1) || C <= 0 || ld < C || act < 0 || act > 3)
        return AVSI_ERR_INVALID_ARG;
    const int has_bn = mean != nullptr;
    if (has_bn && (!rstd || !gamma || !beta)) return AVSI_ERR_INVALID_ARG;
    if ((ld & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(pooled)) & 15))
        return AVSI_ERR_UNSUPPORTED;
    BnArgs a{x, nullptr, mean, rstd, gamma, beta, (int64_t)B * H * W, C, ld, act, has_bn};
    avsi_clear_error();
    hipLaunchKernelGGL((bn_act4_kernel<true>), dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (ld / 4))), dim3(TPB), 0,
                       (hipStream_t)stream, a, y, pooled, B, H, W);
    return avsi_launch_status();
}

// Few-channel convolutions on the 16-wide MFMA (thin_mfma_conv_kernel): (k, C0, C1, Cout) = (3, 16, 32, 16) and (5, 16, 0, 32).
extern "C" int avsi_conv2d_thin_mfma_supported(int k, int C0, int C1, int Cout, int H, int W) {
    return (H % 4 == 0 && W % 32 == 0) && ((k == 3 && C0 == 16 && C1 == 32 && Cout == 16) || (k == 5 && C0 == 16 && C1 == 0 && Cout == 32));
}

// The plain form (no statistics, no deferred batch norm) also takes the two INPUT-GRADIENT convolutions of those layers
// (tap-flipped transposed filters): 32 -> 16, 5 x 5 (dX of the 16 -> 32 encoder layer) and 16 -> 48, 3 x 3 (dX of the decoder
// layer).  As implicit GEMMs they gathered their input once per tap: 6.7 GB for 54 GFLOP.
extern "C" int avsi_conv2d_thin_mfma_plain_supported(int k, int C0, int C1, int Cout, int H, int W) {
    if (avsi_conv2d_thin_mfma_supported(k, C0, C1, Cout, H, W)) return 1;
    return (H % 4 == 0 && W % 32 == 0) && ((k == 5 && C0 == 32 && C1 == 0 && Cout == 16) || (k == 3 && C0 == 16 && C1 == 0 && Cout == 48));
}

static int conv2d_thin_mfma_launch(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H,
                                   int W, int k, const float* filter, int ldf, const float* bias, int Cout, float* out,
                                   int ldo, const float* zeros64, float* part, const float* const* src1_bn, void* stream);
// persistent grid of the 16-wide-MFMA convolution: two workgroups per CU (68 - 71 KB of LDS each), fewer for few tiles
static int thin_mfma_blocks(int n_tiles) { return n_tiles < 2 * AVSI_NUM_CU ? n_tiles : 2 * AVSI_NUM_CU; }

extern "C" int avsi_conv2d_thin_mfma_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H,
                                         int W, int k, const float* filter, int ldf, const float* bias, int Cout, float* out,
                                         int ldo, const float* zeros64, void* stream) {
    return conv2d_thin_mfma_launch(src0, C0, ld0, src1_coarse, C1, ld1, B, H, W, k, filter, ldf, bias, Cout, out, ldo, zeros64,
                                   nullptr, nullptr, stream);
}

static int conv2d_thin_mfma_launch(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H,
                                   int W, int k, const float* filter, int ldf, const float* bias, int Cout, float* out,
                                   int ldo, const float* zeros64, float* part, const float* const* src1_bn, void* stream) {
    if (src1_bn && (k != 3 || !src1_bn[0] || !src1_bn[1] || !src1_bn[2] || !src1_bn[3])) return AVSI_ERR_UNSUPPORTED;
    const Bn4 bn1 = src1_bn ? Bn4{src1_bn[0], src1_bn[1], src1_bn[2], src1_bn[3]} : Bn4{};
    if (!src0 || !filter || !out || !zeros64 || B <= 0 || (C1 && !src1_coarse) || ldf < Cout || ldo < Cout || ld0 < C0 ||
        (C1 && ld1 < C1))
        return AVSI_ERR_INVALID_ARG;
    const bool plain_only = !avsi_conv2d_thin_mfma_supported(k, C0, C1, Cout, H, W);
    if (plain_only && (part || src1_bn)) return AVSI_ERR_UNSUPPORTED;
    if (!avsi_conv2d_thin_mfma_plain_supported(k, C0, C1, Cout, H, W) || (ld0 & 3) || (ld1 & 3) || (ldf & 3) ||
        ((reinterpret_cast<uintptr_t>(src0) | reinterpret_cast<uintptr_t>(src1_coarse) | reinterpret_cast<uintptr_t>(filter) |
          reinterpret_cast<uintptr_t>(zeros64)) & 15))
        return AVSI_ERR_UNSUPPORTED;
    const int n_tiles = B * (H / 4) * (W / 32);
    // (the 32 -> 16, 5 x 5 input gradient holds 89 KB of patch + filter: one workgroup per CU)
    const int blocks = (k == 5 && C0 == 32) ? (n_tiles < AVSI_NUM_CU ? n_tiles : AVSI_NUM_CU) : thin_mfma_blocks(n_tiles);
    const dim3 grid(blocks), block(256);
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
#define AVSI_THIN_MFMA(KS, CA, CB, CO, ST, BN, LDS)                                                                             \
    do {                                                                                                                      \
        (void)hipFuncSetAttribute((const void*)thin_mfma_conv_kernel<KS, CA, CB, CO, ST, BN>,                                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS));                                     \
        hipLaunchKernelGGL((thin_mfma_conv_kernel<KS, CA, CB, CO, ST, BN>), grid, block, (LDS), st, src0, ld0, src1_coarse,    \
                           ld1, filter, ldf, bias, out, ldo, H, W, zeros64, part, bn1, n_tiles);                              \
    } while (0)
    if (k == 5 && C0 == 32) {
        constexpr size_t lds = ((size_t)8 * 36 * 33 + 25 * 32 * 16) * 4;
        AVSI_THIN_MFMA(5, 32, 0, 16, false, false, lds);
    } else if (k == 3 && C1 == 0) {
        constexpr size_t lds = ((size_t)6 * 34 * 17 + 9 * 16 * 48) * 4;
        AVSI_THIN_MFMA(3, 16, 0, 48, false, false, lds);
    } else if (k == 3) {
        constexpr size_t lds = ((size_t)6 * 34 * 49 + 9 * 48 * 16) * 4;
        if (part && src1_bn) AVSI_THIN_MFMA(3, 16, 32, 16, true, true, lds);
        else if (part) AVSI_THIN_MFMA(3, 16, 32, 16, true, false, lds);
        else if (src1_bn) AVSI_THIN_MFMA(3, 16, 32, 16, false, true, lds);
        else AVSI_THIN_MFMA(3, 16, 32, 16, false, false, lds);
    } else {
        constexpr size_t lds = ((size_t)8 * 36 * 17 + 25 * 16 * 32) * 4;
        if (part) AVSI_THIN_MFMA(5, 16, 0, 32, true, false, lds);
        else AVSI_THIN_MFMA(5, 16, 0, 32, false, false, lds);
    }
#undef AVSI_THIN_MFMA
    return avsi_launch_status();
}

// gemm.hip
int avsi_conv2d_stats_parts(int B, int H, int W, int Cout);
int avsi_conv2d_launch(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H, int W, int k,
                       const float* filter, int ldf, const float* bias, int Cout, float* out, int ldo, const float* zeros64,
                       float* stats, void* stream);

// parts rows of `row` floats -> G rows: row g sums rows g, g + G, ... in that order (deterministic).  The per-tile partial
// statistics of a convolution are thousands of rows (16384 for the 64 x 64 layers at 512 clips); the one-wave-per-channel
// finishing kernel walks them with a stride of a whole row -- every access its own cache line, 0.25 ms for those layers --
// so they are folded first, rows read whole and coalesced.
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ part, int parts, int row, float* __restrict__ out) {
    const int G = gridDim.x, g = blockIdx.x;
    const int rpp = 256 / row;                                // rows a pass of the block covers (row <= 256)
    const int e = threadIdx.x % row, rl = threadIdx.x / row;
    __shared__ float red[256];
    float s = 0.f;
    if (rl < rpp)
        for (int p = g + rl * G; p < parts; p += rpp * G) s += part[(int64_t)p * row + e];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < row) {
        float t = 0.f;
        for (int j = 0; j < rpp; ++j) t += red[j * row + threadIdx.x];
        out[(int64_t)g * row + threadIdx.x] = t;
    }
}

// Convolution + the batch statistics of its output (tf.layers.batch_normalization(training=True), unet_layers.py:14,33) in one
// call: the convolution's epilogue leaves per-tile partial sums, one small launch turns them into mean and 1 / sqrt(var + eps)
// -- no pass over the output.  Route: the 16-wide-MFMA kernel where it applies, else the implicit GEMM.
static int conv_bn_parts(int B, int H, int W, int k, int C0, int C1, int Cout) {
    if (avsi_conv2d_thin_mfma_supported(k, C0, C1, Cout, H, W)) return thin_mfma_blocks(B * (H / 4) * (W / 32));
    return avsi_conv2d_stats_parts(B, H, W, Cout);
}
constexpr int FOLD_ROWS = 256;
extern "C" size_t avsi_conv2d_bn_workspace_bytes(int B, int H, int W, int k, int C0, int C1, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    return ((size_t)conv_bn_parts(B, H, W, k, C0, C1, Cout) + FOLD_ROWS) * 2 * (size_t)Cout * sizeof(float);
}
extern "C" int avsi_conv2d_bn_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1,
                                  const float* const* src1_bn, int B, int H, int W, int k, const float* filter, int ldf,
                                  const float* bias, int Cout, float* out, int ldo, const float* zeros64, float eps, float* mean,
                                  float* rstd, void* workspace, size_t workspace_bytes, void* stream) {
    if (!mean || !rstd || B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return AVSI_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < avsi_conv2d_bn_workspace_bytes(B, H, W, k, C0, C1, Cout)) return AVSI_ERR_WORKSPACE;
    float* part = static_cast<float*>(workspace);
    const int parts = conv_bn_parts(B, H, W, k, C0, C1, Cout);
    int rc;
    if (avsi_conv2d_thin_mfma_supported(k, C0, C1, Cout, H, W))
        rc = conv2d_thin_mfma_launch(src0, C0, ld0, src1_coarse, C1, ld1, B, H, W, k, filter, ldf, bias, Cout, out, ldo, zeros64, part,
                                     src1_bn, stream);
    else if (src1_bn)
        return AVSI_ERR_UNSUPPORTED;          // the implicit GEMM gathers its operand by DMA: nothing can be applied on the way
    else
        rc = avsi_conv2d_launch(src0, C0, ld0, src1_coarse, C1, ld1, B, H, W, k, filter, ldf, bias, Cout, out, ldo, zeros64, part,
                                stream);
    if (rc != AVSI_OK) return rc;
    const float* fin = part;
    int nfin = parts;
    if (parts > 4 * FOLD_ROWS && 2 * Cout <= 256) {
        float* folded = part + (size_t)parts * 2 * Cout;
        hipLaunchKernelGGL(fold_partials_kernel, dim3(FOLD_ROWS), dim3(256), 0, (hipStream_t)stream, (const float*)part, parts, 2 * Cout,
                           folded);
        fin = folded, nfin = FOLD_ROWS;
    }
    hipLaunchKernelGGL(colpair_final_kernel<0>, dim3(Cout), dim3(64), 0, (hipStream_t)stream, fin, nfin, Cout, (int64_t)B * H * W, eps,
                       mean, rstd);
    return avsi_launch_status();
}

extern "C" int avsi_conv2d_thin_relu_pool_f32(const float* src0, int ld0, int B, int H, int W, int k, const float* filter, int ldf,
                                              const float* bias, int Cout, float* out, int ldo, void* stream) {
    if (!src0 || !filter || !out || B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1) || ld0 < 1 || ldf < Cout || ldo < Cout)
        return AVSI_ERR_INVALID_ARG;
    if (k != 7 || Cout != 16 || (ldo & 3) || (reinterpret_cast<uintptr_t>(out) & 15)) return AVSI_ERR_UNSUPPORTED;
    avsi_clear_error();
    if (H % 16 == 0 && W % 64 == 0)
        hipLaunchKernelGGL((conv7_c1_mfma_kernel<true>), dim3(B * (H / 16) * (W / 64)), dim3(256), 0, (hipStream_t)stream, src0, ld0,
                           filter, ldf, bias, out, ldo, H, W);
    else
        hipLaunchKernelGGL((direct_conv_relu_pool_kernel<7, 16>), dim3(grid_for((int64_t)B * (H / 2) * (W / 2))), dim3(TPB), 0,
                           (hipStream_t)stream, src0, ld0, filter, ldf, bias, out, ldo, B, H, W);
    return avsi_launch_status();
}

extern "C" int avsi_bn_act_bwd_f32(const float* x, const float* dy, int64_t R, int C, int ld, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, int act, float* dx,
                                   float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dx || R <= 0 || C <= 0 || ld < C || act < 0 || act > 3) return AVSI_ERR_INVALID_ARG;
    const int has_bn = mean != nullptr;
    if (has_bn && (!rstd || !gamma || !beta || !dgamma || !dbeta)) return AVSI_ERR_INVALID_ARG;
    if (has_bn && (!workspace || workspace_bytes < avsi_unet_workspace_bytes(C))) return AVSI_ERR_WORKSPACE;
    if (has_bn && ((ld & 3) || ld > 256 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15)))
        return AVSI_ERR_UNSUPPORTED;
    BnArgs a{x, dy, mean, rstd, gamma, beta, R, C, ld, act, has_bn};
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    if (has_bn) {
        const int parts = parts_for(R);
        hipLaunchKernelGGL(colpair_partial_kernel<1>, dim3(1, parts), dim3(TPB), 0, st, a,
                           (float*)workspace);
        hipLaunchKernelGGL(colpair_final_kernel<1>, dim3(C), dim3(64), 0, st,
                           (const float*)workspace, parts, C, R, 0.f, dbeta, dgamma);
    }
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(grid_for(R * ld)), dim3(TPB), 0, st, a, dbeta, dgamma, dx);
    return avsi_launch_status();
}

extern "C" int avsi_bn_act_pool_bwd_f32(const float* x, const float* dpooled, int B, int H, int W, int C, int ld,
                                        const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                                        float* dx, float* dgamma, float* dbeta, float* dbias, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (!x || !dpooled || !dx || B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1) || C <= 0 || ld < C || act < 0 || act > 3)
        return AVSI_ERR_INVALID_ARG;
    const int has_bn = mean != nullptr;
    if (has_bn && (!rstd || !gamma || !beta || !dgamma || !dbeta)) return AVSI_ERR_INVALID_ARG;
    if ((ld & 3) || ld > 256 ||
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dpooled) | reinterpret_cast<uintptr_t>(dx)) & 15))
        return AVSI_ERR_UNSUPPORTED;
    if ((has_bn || dbias) && (!workspace || workspace_bytes < avsi_unet_workspace_bytes(C))) return AVSI_ERR_WORKSPACE;
    const int64_t R = (int64_t)B * H * W;
    BnArgs a{x, dpooled, mean, rstd, gamma, beta, R, C, ld, act, has_bn};
    const hipStream_t st = (hipStream_t)stream;
    // row slabs: a workgroup passes over TPB / (ld / 4) pooled pixels at a time; four passes each at least, so that the deep
    // layers (2048 pooled pixels of 128 channels: 4 workgroups by parts_for) still spread over the chip
    const int64_t per_pass = TPB / (ld >> 2);
    int64_t want = avsi_ceil_div(R / 4, per_pass * 4);
    const int parts = (int)(want < 1 ? 1 : (want > MAXPARTS ? MAXPARTS : want));
    float* part = (float*)workspace;
    avsi_clear_error();
    if (has_bn) {
        hipLaunchKernelGGL(bn_act_pool_bwd_kernel<false>, dim3(1, parts), dim3(TPB), 0, st, a, B, H, W, (const float*)nullptr,
                           (const float*)nullptr, (float*)nullptr, part);
        hipLaunchKernelGGL(colpair_final_kernel<1>, dim3(C), dim3(64), 0, st, (const float*)part, parts, C, R, 0.f, dbeta, dgamma);
    }
    hipLaunchKernelGGL(bn_act_pool_bwd_kernel<true>, dim3(1, parts), dim3(TPB), 0, st, a, B, H, W, (const float*)dbeta,
                       (const float*)dgamma, dx, dbias ? part : (float*)nullptr);
    if (dbias)   // column sums of dx; the second row of the pairs is zero and lands behind the first in the workspace
        hipLaunchKernelGGL(colpair_final_kernel<1>, dim3(C), dim3(64), 0, st, (const float*)part, parts, C, R, 0.f, dbias,
                           part + (size_t)parts * 2 * C);
    return avsi_launch_status();
}

extern "C" int avsi_maxpool2_f32(const float* x, float* y, int B, int H, int W, int C, int ld, void* stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1) || C <= 0 || ld < C) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * ld)), dim3(TPB), 0,
                       (hipStream_t)stream, x, y, B, H, W, C, ld);
    return avsi_launch_status();
}

extern "C" int avsi_maxpool2_bwd_f32(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int ld,
                                     void* stream) {
    if (!x || !dy || !dx || B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1) || C <= 0 || ld < C) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * ld)), dim3(TPB), 0,
                       (hipStream_t)stream, x, dy, dx, B, H, W, C, ld);
    return avsi_launch_status();
}
