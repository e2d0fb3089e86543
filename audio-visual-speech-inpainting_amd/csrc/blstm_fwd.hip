// Recurrent half of one bidirectional LSTM layer on gfx950: for every step
//     z_t = xproj_t + h_{t-1} . Wh ;  (i, j, f, o) = split(z_t)
//     c_t = sigmoid(f) c_{t-1} + sigmoid(i) tanh(j) ;  h_t = sigmoid(o) tanh(c_t)
// which is CudnnCompatibleLSTMCell / LSTMBlockCell(forget_bias = 0) under
// stack_bidirectional_dynamic_rnn (reference models.py:106-115).  The
// input half x_t . Wx + b of the cell's [x_t, h_{t-1}] . K product is time-independent and is
// hoisted into one large GEMM (gemm.hip) that produces `xproj`.
//
// Mapping (batch-stationary, no inter-workgroup traffic):
//   - a workgroup owns MT x 32 utterances of one direction for all T steps; h_{t-1} of those
//     utterances lives in LDS (double buffered), c_t in registers; nothing is exchanged between
//     workgroups, so there is no grid barrier and no cross-XCD visibility protocol at all;
//   - 8 waves; wave w owns hidden units [32w, 32w+32) and ALL FOUR gates of those units, i.e.
//     4 MFMA column tiles x MT row tiles of v_mfma_f32_32x32x2_f32 (exact fp32).  The gate
//     pre-activations of one (utterance, unit) therefore meet in ONE lane at the same register
//     index, and the whole cell update is lane-local: no shuffles, no LDS in the epilogue;
//   - Wh (padded 256 x 1024, 1 MiB per direction) is streamed from the XCD's L2 every step in a
//     host-packed fragment order: one coalesced 16-byte load per lane feeds 4 k-steps;
//   - h_{t-1} fragments come from LDS with ds_read_b128 under the same k-permutation as gemm.hip.
// Layouts (floats):
//   xproj [T][Bp][2][1024]   packed gate columns: col = 128 w + 32 gate + unit % 32, unit = 32 w + ..
//   whp   [2][8 w][32 q][4 gate][64 lane][4 s] = Wh[k = 8q + 4(lane>>5) + s][unit 32w + (lane&31)][gate]
//   hout  [T][Bp][512]       fw units at 0..255, bw units at 256..511 (units >= H stay exactly 0)
//   resv  [T][Bp][2][5][256] (training only) i, j, f, o after activation, then c_t
#include "avsi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HP = 256;         // padded hidden size
constexpr int HS = HP + 4;      // LDS row stride of the h tile (bank-conflict-free b128 reads)
constexpr int NWAVE = 8;
constexpr int GP = 4 * HP;      // packed gate columns per direction

struct RecArgs {
    const float* xproj;
    const float* whp;
    float* hout;
    float* resv;
    int T, Bp;
    unsigned long long* stamps;  // diagnostic builds only (-DAVSI_REC_STAMPS): [wg][wave][4] phase cycles
    int row0, rend;              // the utterances [row0, rend) of the batch are this launch's (the whole batch: 0, Bp)
    int diag;                    // AVSI_REC_Q_DIAG (quarter-product kernel, timing experiments; results WRONG): see launch_rec_q
};

// In-kernel phase stamps (diagnostic build only; the shipped library never executes one).
#ifdef AVSI_REC_STAMPS
#define AVSI_STAMP(var)                                                                  \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define AVSI_STAMP(var) do { } while (0)
#endif

__device__ __forceinline__ float sigmoidf_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_fast(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

// One 8-wide k group: 4 MFMA k-steps x MT row tiles x 4 gate tiles.  `af` holds this group's
// h_{t-1} fragments; the NEXT group's fragments are fetched into `an` first and pinned above the
// MFMAs (sched_barrier) so their LDS latency is covered by this group's 32 MFMAs.
template <int MT>
__device__ __forceinline__ void read_h(float4 (&af)[MT], const float* __restrict__ hcur, int q, int li, int hi) {
#pragma unroll
    for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const float4*>(hcur + (m * 32 + li) * HS + 8 * q + 4 * hi);
}

template <int MT>
__device__ __forceinline__ void kgroup(f32x16 (&acc)[MT][4], const float4 (&b)[4], const float4 (&af)[MT]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float av = s == 0 ? af[m].x : s == 1 ? af[m].y : s == 2 ? af[m].z : af[m].w;
                const float bv = s == 0 ? b[g].x : s == 1 ? b[g].y : s == 2 ? b[g].z : b[g].w;
                acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][g], 0, 0, 0);
            }
}

// Buffer-resource helpers: wave-uniform base in SGPRs + one 32-bit lane offset + a scalar offset
// per access.  Keeps the 128+ per-step row addresses out of the VGPR file (they are scalar
// immediates) and gives the batch tail for free: rows past num_records read 0 / drop stores.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
// the BPTT reserve is written once and read once, a step of the backward pass later: `nt` keeps it from displacing Wh in L2
#ifndef AVSI_RESV_AUX
#define AVSI_RESV_AUX 2
#endif
#ifndef AVSI_XPROJ_AUX
#define AVSI_XPROJ_AUX 0
#endif
__device__ __forceinline__ float buf_load_x(rsrc_t r, int voff, int soff) {      // the gate pre-activations: read once
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AVSI_XPROJ_AUX));
}
__device__ __forceinline__ void buf_store_nt(rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, AVSI_RESV_AUX);
}
// MT = 1 (32 rows) is built for TWO workgroups per CU (<= 128 VGPRs, 66.5 KB LDS each): the
// two run out of phase, so one's xproj loads / cell epilogue / barrier hide under the other's MFMAs.
template <int MT, bool SAVE>
__global__ __launch_bounds__(512, MT == 1 ? 4 : 2) void blstm_rec_fwd_kernel(const RecArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hbuf = reinterpret_cast<float*>(smem);  // [2][MT*32][HS]
    constexpr int HTILE = MT * 32 * HS;
    constexpr int XROW = 2 * GP * 4, HROW = 2 * HP * 4, RROW = 2 * 5 * HP * 4;  // row pitches, bytes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
    const int li = lane & 31, hi = lane >> 5;
    const int dir = blockIdx.y;
    const int b0 = a.row0 + blockIdx.x * (MT * 32);
    const int T = a.T, Bp = a.Bp;
    const int live_rows = min(MT * 32, a.rend - b0);

    for (int i = tid; i < 2 * HTILE; i += 512) hbuf[i] = 0.f;

    // Wh fragments: plain 16-byte global loads (uniform base + lane).  NB the b128 raw-buffer
    // builtin of this toolchain lowers to a single dword load, so it is not used.
    const float4* __restrict__ wp =
        reinterpret_cast<const float4*>(a.whp) + (size_t)(dir * NWAVE + w) * (32 * 4 * 64) + lane;
    const int voff_x = 4 * hi * XROW + (dir * GP + w * 128 + li) * 4;
    const int voff_h = 4 * hi * HROW + (dir * HP + w * 32 + li) * 4;
    const int voff_r = 4 * hi * RROW + (dir * 5 * HP + w * 32 + li) * 4;

    f32x16 c[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[m][r] = 0.f;

    __syncthreads();

    unsigned long long ph_load = 0, ph_mfma = 0, ph_cell = 0, ph_bar = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    (void)ph_load, (void)ph_mfma, (void)ph_cell, (void)ph_bar, (void)s0, (void)s1, (void)s2, (void)s3, (void)s4;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        AVSI_STAMP(s0);
        const float* hcur = hbuf + (step & 1) * HTILE;
        float* hnext = hbuf + ((step & 1) ^ 1) * HTILE;
        const size_t row0 = (size_t)t * Bp + b0;
        const rsrc_t rx = make_rsrc(a.xproj + row0 * (2 * GP), live_rows * XROW);
        const rsrc_t rh = make_rsrc(a.hout + row0 * (2 * HP), live_rows * HROW);

        // ---- accumulators start from the hoisted input projection (bias already inside)
        f32x16 acc[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[m][g][r] = buf_load_x(rx, voff_x, (m * 32 + (r & 3) + 8 * (r >> 2)) * XROW + g * 128);

        // ---- z += h_{t-1} . Wh : 32 k-groups of 8, Wh fragments prefetched one group ahead.
        //      Two named register sets (static indexing) ping-pong; see kgroup().
#ifdef AVSI_REC_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // diagnostic: isolate the xproj load phase
#endif
        AVSI_STAMP(s1);
        float4 bw0[4], bw1[4], af0[MT], af1[MT];
#pragma unroll
        for (int g = 0; g < 4; ++g) bw0[g] = wp[g * 64];
        read_h<MT>(af0, hcur, 0, li, hi);
        for (int q = 0; q < 32; q += 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bw1[g] = wp[((q + 1) * 4 + g) * 64];
            read_h<MT>(af1, hcur, q + 1, li, hi);
            __builtin_amdgcn_sched_barrier(0);
            kgroup<MT>(acc, bw0, af0);
            if (q + 2 < 32) {
#pragma unroll
                for (int g = 0; g < 4; ++g) bw0[g] = wp[((q + 2) * 4 + g) * 64];
                read_h<MT>(af0, hcur, q + 2, li, hi);
            }
            __builtin_amdgcn_sched_barrier(0);
            kgroup<MT>(acc, bw1, af1);
        }

        AVSI_STAMP(s2);
        // ---- LSTM cell, lane-local: (row, unit) = (C/D row of register r, 32 w + li)
        rsrc_t rr = rh;
        if (SAVE) rr = make_rsrc(a.resv + row0 * (2 * 5 * HP), live_rows * RROW);
        float* hn_lds = hnext + (4 * hi) * HS + w * 32 + li;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowc = m * 32 + (r & 3) + 8 * (r >> 2);
                const float ig = sigmoidf_fast(acc[m][0][r]);
                const float jg = tanhf_fast(acc[m][1][r]);
                const float fg = sigmoidf_fast(acc[m][2][r]);
                const float og = sigmoidf_fast(acc[m][3][r]);
                const float cn = fg * c[m][r] + ig * jg;
                c[m][r] = cn;
                const float hn = og * tanhf_fast(cn);
                hn_lds[rowc * HS] = hn;
                buf_store(rh, voff_h, rowc * HROW, hn);
                if (SAVE) {
                    buf_store_nt(rr, voff_r, rowc * RROW + 0 * HP * 4, ig);
                    buf_store_nt(rr, voff_r, rowc * RROW + 1 * HP * 4, jg);
                    buf_store_nt(rr, voff_r, rowc * RROW + 2 * HP * 4, fg);
                    buf_store_nt(rr, voff_r, rowc * RROW + 3 * HP * 4, og);
                    buf_store_nt(rr, voff_r, rowc * RROW + 4 * HP * 4, cn);
                }
            }
        }
        AVSI_STAMP(s3);
        __syncthreads();
        AVSI_STAMP(s4);
#ifdef AVSI_REC_STAMPS
        ph_load += s1 - s0, ph_mfma += s2 - s1, ph_cell += s3 - s2, ph_bar += s4 - s3;
#endif
    }
#ifdef AVSI_REC_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NWAVE + w) * 4;
        o[0] = ph_load, o[1] = ph_mfma, o[2] = ph_cell, o[3] = ph_bar;
    }
#endif
}

// ------------------------------------------------------------------------------------------
// Ping-pong variant (64 rows per workgroup as two 32-row tiles that alternate roles).
// The straightforward kernel above spends ~22 % of every step outside the MFMA loop (xproj
// loads 5 %, cell epilogue 12 %, barrier 5 % -- measured with tools/rec_stamps.cpp), and all 8
// waves are in those phases at the same time, so the matrix pipe idles.  Here each step is two
// phases separated by barriers; in a phase ONE tile issues its 512 MFMAs while, in the same
// instruction stream, the OTHER tile runs its cell update for the step it just finished and
// then starts the loads of its next gate pre-activations straight into its (now free)
// accumulator registers:
//     phase B(s): MFMA tile 1, step s   ||  cell tile 0, step s   ->  load tile 0, step s+1
//     phase A(s): MFMA tile 0, step s+1 ||  cell tile 1, step s   ->  load tile 1, step s+1
// A tile's h is written in one phase and read in the other, so one LDS buffer per tile suffices
// (66.5 KB).  Wh is streamed from L2 twice per step (2 MiB per workgroup-step).
// ------------------------------------------------------------------------------------------
// Wave-uniform global pointer made opaque to the optimiser, so that every load through it is
// "SGPR base + shared lane VGPR + immediate" (global_load_dwordx4 v, v_lane, s[..] offset:imm).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f* gptr4;
__device__ __forceinline__ gptr4 opaque_base(const float4* p) {
    gptr4 g = (gptr4)(const void*)p;
    asm volatile("" : "+s"(g));
    return g;
}
// (a running pointer, opaque after every advance: the 32 bases `wb + q * 256` of a phase are loop-invariant, the compiler
// computed them ahead of the time loop, parked them in VGPR lanes and fetched each back with two v_readlane per group)
__device__ __forceinline__ gptr4 opaque_next(gptr4 g, int float4s) {
    g += float4s;
    asm volatile("" : "+s"(g));
    return g;
}
__device__ __forceinline__ float4 ldg4(gptr4 p, int idx) {
    const v4f v = p[idx];
    return make_float4(v.x, v.y, v.z, v.w);
}

template <int X, bool DO_MFMA, bool SAVE, bool AVSI_PP_RESIDENT0>
__device__ __forceinline__ void pp_phase(f32x16 (&acc)[2][4], f32x16 (&c)[2], float* __restrict__ hbuf,
                                         const float4* __restrict__ wb, const float4 (&bfirst)[4], const int lane, const rsrc_t rx_next,
                                         const rsrc_t rh, const rsrc_t rr, const int voff_x, const int voff_h,
                                         const int voff_r, const int w, const int li, const int hi) {
    constexpr int Y = 1 - X;
    constexpr int XROW = 2 * GP * 4, HROW = 2 * HP * 4, RROW = 2 * 5 * HP * 4;
    const float* hx = hbuf + X * (32 * HS);                                 // tile X: read by the MFMAs
    float* hy = hbuf + Y * (32 * HS) + (4 * hi) * HS + w * 32 + li;         // tile Y: written by the cell
    float4 bw[2][4], af[2];
    f32x16 nx[4];  // tile Y's next pre-activations: loaded once acc[Y] is dead, then become acc[Y]
    // Every phase starts with the SAME fragment group (k = 0 .. 7 of this wave's columns): it stays in registers for the
    // whole launch (AVSI_PP_RESIDENT0, round 6) -- a phase used to open with these four loads right behind its barrier and
    // nothing to do until they came back from L2 (twice per step: ~1 us of a 64 us step)
    gptr4 wrun = opaque_base(wb + (AVSI_PP_RESIDENT0 ? 256 : 0));                // the fragment group requested next
    if (DO_MFMA) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bw[0][g] = AVSI_PP_RESIDENT0 ? bfirst[g] : ldg4(wrun, g * 64 + lane);
        if (!AVSI_PP_RESIDENT0) wrun = opaque_next(wrun, 256);
        af[0] = *reinterpret_cast<const float4*>(hx + li * HS + 4 * hi);
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int cb = q & 1;
        if (DO_MFMA) {
            if (q + 1 < 32) {
                // Without the opaque base hipcc materialises one 64-bit VGPR address per load (128 of
                // them), hoists them out of the step loop and spills them.
#pragma unroll
                for (int g = 0; g < 4; ++g) bw[cb ^ 1][g] = ldg4(wrun, g * 64 + lane);
                wrun = opaque_next(wrun, 256);
                af[cb ^ 1] = *reinterpret_cast<const float4*>(hx + li * HS + 8 * (q + 1) + 4 * hi);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float av = s == 0 ? af[cb].x : s == 1 ? af[cb].y : s == 2 ? af[cb].z : af[cb].w;
                    const float4 b4 = bw[cb][g];
                    const float bv = s == 0 ? b4.x : s == 1 ? b4.y : s == 2 ? b4.z : b4.w;
                    acc[X][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[X][g], 0, 0, 0);
                }
        }
        if (q < 16) {  // one (row-register) element of tile Y per k group, in the first half of the phase
            const int r = q;
            const int rowc = Y * 32 + (r & 3) + 8 * (r >> 2);
            const float ig = sigmoidf_fast(acc[Y][0][r]);
            const float jg = tanhf_fast(acc[Y][1][r]);
            const float fg = sigmoidf_fast(acc[Y][2][r]);
            const float og = sigmoidf_fast(acc[Y][3][r]);
            const float cn = fg * c[Y][r] + ig * jg;
            c[Y][r] = cn;
            const float hn = og * tanhf_fast(cn);
            hy[((r & 3) + 8 * (r >> 2)) * HS] = hn;
            buf_store(rh, voff_h, rowc * HROW, hn);
            if (SAVE) {
                buf_store_nt(rr, voff_r, rowc * RROW + 0 * HP * 4, ig);
                buf_store_nt(rr, voff_r, rowc * RROW + 1 * HP * 4, jg);
                buf_store_nt(rr, voff_r, rowc * RROW + 2 * HP * 4, fg);
                buf_store_nt(rr, voff_r, rowc * RROW + 3 * HP * 4, og);
                buf_store_nt(rr, voff_r, rowc * RROW + 4 * HP * 4, cn);
            }
        } else if (q < 24) {
            // second half: acc[Y] is dead, its registers take the next step's pre-activations
            // (2 row-registers per k group; a zero-record descriptor makes these loads return 0)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * (q - 16) + e;
                const int rowc = Y * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int g = 0; g < 4; ++g) nx[g][r] = buf_load_x(rx_next, voff_x, rowc * XROW + g * 128);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[Y][g] = nx[g];
}

template <bool SAVE, bool AVSI_PP_RESIDENT0 = true>
__global__ __launch_bounds__(512, 2) void blstm_rec_fwd_pp_kernel(const RecArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hbuf = reinterpret_cast<float*>(smem);  // [2 tiles][32][HS]
    constexpr int XROW = 2 * GP * 4, HROW = 2 * HP * 4, RROW = 2 * 5 * HP * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int dir = blockIdx.y;
    const int b0 = a.row0 + blockIdx.x * 64;
    const int T = a.T, Bp = a.Bp;
    const int live_rows = min(64, a.rend - b0);

    for (int i = tid; i < 2 * 32 * HS; i += 512) hbuf[i] = 0.f;

    // wave-uniform base (SGPR pair) + lane: every fragment address is scalar base + one shared VGPR
    const float4* __restrict__ wb = reinterpret_cast<const float4*>(a.whp) + (size_t)(dir * NWAVE + w) * (32 * 4 * 64);
    const int voff_x = 4 * hi * XROW + (dir * GP + w * 128 + li) * 4;
    const int voff_h = 4 * hi * HROW + (dir * HP + w * 32 + li) * 4;
    const int voff_r = 4 * hi * RROW + (dir * 5 * HP + w * 32 + li) * 4;

    f32x16 acc[2][4], c[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[m][r] = 0.f;
    float4 bfirst[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bfirst[g] = AVSI_PP_RESIDENT0 ? wb[g * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);

    auto row0_of = [&](int step) { return ((size_t)(dir ? (T - 1 - step) : step)) * Bp + b0; };
    {   // gate pre-activations of step 0 for both tiles
        const rsrc_t rx = make_rsrc(a.xproj + row0_of(0) * (2 * GP), live_rows * XROW);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[m][g][r] = buf_load_x(rx, voff_x, (m * 32 + (r & 3) + 8 * (r >> 2)) * XROW + g * 128);
    }
    __syncthreads();
    auto out_rsrc = [&](int step, rsrc_t& rh, rsrc_t& rr) {
        const size_t r = row0_of(step);
        rh = make_rsrc(a.hout + r * (2 * HP), live_rows * HROW);
        rr = SAVE ? make_rsrc(a.resv + r * (2 * 5 * HP), live_rows * RROW) : rh;
    };
    // next-step pre-activations; `on` = false gives a zero-record descriptor (loads return 0, no branch)
    auto in_rsrc = [&](int step, bool on) {
        return make_rsrc(a.xproj + row0_of(step) * (2 * GP), on ? live_rows * XROW : 0);
    };
    rsrc_t rh, rr;
    // prologue: h_{-1} = 0, so the step-0 pre-activations are final without any MFMA.
    //   cell tile 0 (step 0) -> load tile 0 (step 1)
    out_rsrc(0, rh, rr);
    pp_phase<1, false, SAVE, AVSI_PP_RESIDENT0>(acc, c, hbuf, wb, bfirst, lane, in_rsrc(T > 1 ? 1 : 0, T > 1), rh, rr, voff_x, voff_h, voff_r, w, li, hi);
    AVSI_LDS_BARRIER();
    for (int step = 0; step + 1 < T; ++step) {
        // phase A: MFMA tile 0 (step + 1) || cell tile 1 (step) -> load tile 1 (step + 1)
        out_rsrc(step, rh, rr);
        pp_phase<0, true, SAVE, AVSI_PP_RESIDENT0>(acc, c, hbuf, wb, bfirst, lane, in_rsrc(step + 1, true), rh, rr, voff_x, voff_h, voff_r, w, li, hi);
        AVSI_LDS_BARRIER();
        // phase B: MFMA tile 1 (step + 1) || cell tile 0 (step + 1) -> load tile 0 (step + 2)
        const bool more = step + 2 < T;
        out_rsrc(step + 1, rh, rr);
        pp_phase<1, true, SAVE, AVSI_PP_RESIDENT0>(acc, c, hbuf, wb, bfirst, lane, in_rsrc(more ? step + 2 : step + 1, more), rh, rr, voff_x, voff_h, voff_r,
                                w, li, hi);
        AVSI_LDS_BARRIER();
    }
    // epilogue: cell tile 1 (step T - 1)
    out_rsrc(T - 1, rh, rr);
    pp_phase<0, false, SAVE, AVSI_PP_RESIDENT0>(acc, c, hbuf, wb, bfirst, lane, in_rsrc(T - 1, false), rh, rr, voff_x, voff_h, voff_r, w, li, hi);
}

// ------------------------------------------------------------------------------------------
// Quarter-block variant (round 6; 64 rows per workgroup, Wh fetched from L2 ONCE per step and workgroup).
// The ping-pong kernel above hides a tile's cell under the other tile's MFMAs, and pays for it with the weight stream:
// its two tiles are in opposite phases, so every fragment of Wh comes up from L2 twice per step (128 of the 159 GB a
// launch moves between L2 and L1; the L1's miss queue is a third full and the socket's power cap takes 4 % of the clock,
// DESIGN 4.3).  The kernel that puts both tiles into ONE MFMA phase (blstm_rec_fwd_kernel<2>) halves the stream and
// runs at 2.38 instead of 2.21 GHz -- but its cells run in the open (matrix pipes busy 0.72 instead of 0.89: 19.1 ms
// against 16.3).  Here both are had: the split that makes room for the cell is over the HIDDEN UNITS, not over the rows.
//   - wave w = (tile, v): row tile w / 4, unit blocks 2 v (set U0) and 2 v + 1 (set U1) -- 64 units x 4 gates, one row tile:
//     the same 128 accumulator registers as a ping-pong wave.  Waves w and w + 4 share a SIMD and read the SAME
//     fragments at the same time: one of the two requests misses in the CU's L1, the other is merged with it;
//   - a step is four quarter products Q[cols][k set] of 256 MFMAs per wave, ordered so that every cell has matrix work
//     of OTHER columns beside it, on h values that exist already:
//         Q00(s)  U0 columns, k in U0   ||  cell U1 (step s - 1) -> h_{s-1}[U1], then loads x_s[U1]
//         -- barrier --
//         Q01(s)  U0 columns, k in U1                              (z_s[U0] complete)
//         Q11(s)  U1 columns, k in U1   ||  cell U0 (step s)     -> h_s[U0] into the OTHER h buffer
//         Q10(s)  U1 columns, k in U0   ||  loads x_{s+1}[U0]
//         -- barrier --
//     h is double buffered (133 KB: one workgroup per CU, as the ping-pong kernel at this batch size);
//   - measured (profiles/r06_rec_fwd_q_pmc.txt, r06_rec_fwd_q_hot.txt): as fast as the ping-pong kernel, no faster -- the L1 does
//     not merge the pair's requests, and even a third fewer L2 requests (AVSI_REC_Q_DIAG=1: the second tile's waves read one hot
//     group; results wrong) change neither the launch time nor the clock.  Kept as the measured alternative (rows_per_wg = 66);
//   - sums: x + (k in U0) + (k in U1) for U0 columns, x + (k in U1) + (k in U0) for U1 columns -- another order than
//     the other kernels' 0 .. 255: equal to rounding, not to the bit.
// ------------------------------------------------------------------------------------------
// MODE: 0 = matrix work only, 1 = + the cell of the OTHER unit set (one element per k group), 2 = + loads of its next
// pre-activations (two elements per k group, groups 0 .. 7), 3 = cell and loads (loads in groups 8 .. 15: behind their cells)
template <int KSET, int MODE, bool DO_MFMA, bool SAVE>
__device__ __forceinline__ void q_block(f32x16 (&am)[4], const float* __restrict__ hx, const float4* __restrict__ wb, const int lane, const bool hot,
                                        f32x16 (&ac)[4], f32x16& cc, float* __restrict__ hy, const rsrc_t rh, const rsrc_t rr,
                                        const rsrc_t rx_next, const int voff_h, const int voff_r, const int voff_x) {
    constexpr int XROW = 2 * GP * 4, HROW = 2 * HP * 4, RROW = 2 * 5 * HP * 4;
    float4 bw[2][4], af[2];
    gptr4 wrun = opaque_base(wb + KSET * 4 * 256);       // the fragment group requested next: k groups KSET * 4 + {0..3, 8..11, ..}
    if (DO_MFMA) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bw[0][g] = ldg4(wrun, g * 64 + lane);
        wrun = opaque_next(wrun, hot ? 0 : 256);
        af[0] = *reinterpret_cast<const float4*>(hx + 8 * (KSET * 4));
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int cb = j & 1;
        if (DO_MFMA) {
            if (j + 1 < 16) {
                const int qn = ((j + 1) >> 2) * 8 + KSET * 4 + ((j + 1) & 3);
                if (((j + 1) & 3) == 0) wrun = opaque_next(wrun, hot ? 0 : 4 * 256);     // over the other set's four groups
#pragma unroll
                for (int g = 0; g < 4; ++g) bw[cb ^ 1][g] = ldg4(wrun, g * 64 + lane);
                wrun = opaque_next(wrun, hot ? 0 : 256);
                af[cb ^ 1] = *reinterpret_cast<const float4*>(hx + 8 * qn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float av = s == 0 ? af[cb].x : s == 1 ? af[cb].y : s == 2 ? af[cb].z : af[cb].w;
                    const float4 b4 = bw[cb][g];
                    const float bv = s == 0 ? b4.x : s == 1 ? b4.y : s == 2 ? b4.z : b4.w;
                    am[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, am[g], 0, 0, 0);
                }
        }
        if (MODE & 1) {
            const int r = j;
            const int rowc = (r & 3) + 8 * (r >> 2);
            const float ig = sigmoidf_fast(ac[0][r]);
            const float jg = tanhf_fast(ac[1][r]);
            const float fg = sigmoidf_fast(ac[2][r]);
            const float og = sigmoidf_fast(ac[3][r]);
            const float cn = fg * cc[r] + ig * jg;
            cc[r] = cn;
            const float hn = og * tanhf_fast(cn);
            hy[rowc * HS] = hn;
            buf_store(rh, voff_h, rowc * HROW, hn);
            if (SAVE) {
                buf_store_nt(rr, voff_r, rowc * RROW + 0 * HP * 4, ig);
                buf_store_nt(rr, voff_r, rowc * RROW + 1 * HP * 4, jg);
                buf_store_nt(rr, voff_r, rowc * RROW + 2 * HP * 4, fg);
                buf_store_nt(rr, voff_r, rowc * RROW + 3 * HP * 4, og);
                buf_store_nt(rr, voff_r, rowc * RROW + 4 * HP * 4, cn);
            }
        }
        if ((MODE & 2) && (MODE == 2 ? j < 8 : j >= 8)) {
            // the cell's accumulators are dead from its element on: they take the next step's pre-activations
            // (a zero-record descriptor makes these loads return 0)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * (j & 7) + e;
                const int rowc = (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int g = 0; g < 4; ++g) ac[g][r] = buf_load_x(rx_next, voff_x, rowc * XROW + g * 128);
            }
        }
    }
}

template <bool SAVE>
__global__ __launch_bounds__(512, 2) void blstm_rec_fwd_q_kernel(const RecArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hbuf = reinterpret_cast<float*>(smem);  // [2 buffers][64 rows][HS]
    constexpr int XROW = 2 * GP * 4, HROW = 2 * HP * 4, RROW = 2 * 5 * HP * 4;
    constexpr int HBUF = 64 * HS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = w & 3, trow = (w >> 2) * 32;
    const int li = lane & 31, hi = lane >> 5;
    const int dir = blockIdx.y;
    const int b0 = a.row0 + blockIdx.x * 64;
    const int T = a.T, Bp = a.Bp;
    // (the wave's row tile is part of the descriptors' base: row offsets stay compile-time constants, as in the kernels above;
    //  with the tile's first row as a run-time term every one of the 64 + 64 + 16 offsets of a step became a live SGPR: 199 spilled)
    // (readfirstlane: min / max of uniform values is matched to v_med3_i32, the descriptors then live in VGPRs and every buffer
    //  access is wrapped in a waterfall loop -- 700 branches in this kernel)
    const int live_rows = __builtin_amdgcn_readfirstlane(max(0, min(32, a.rend - b0 - trow)));

    for (int i = tid; i < 2 * HBUF; i += 512) hbuf[i] = 0.f;

    // the two unit blocks of this wave: fragments, and the lane's columns of xproj / hout / reserve
    const float4* __restrict__ wb0 = reinterpret_cast<const float4*>(a.whp) + (size_t)(dir * NWAVE + 2 * v) * (32 * 4 * 64);
    const float4* __restrict__ wb1 = wb0 + 32 * 4 * 64;
    const bool diag_hot = (a.diag & 1) && trow != 0;      // (experiment: the second tile's waves read ONE fragment group over and over)
    const int voff_x0 = 4 * hi * XROW + (dir * GP + (2 * v) * 128 + li) * 4, voff_x1 = voff_x0 + 128 * 4;
    const int voff_h0 = 4 * hi * HROW + (dir * HP + (2 * v) * 32 + li) * 4, voff_h1 = voff_h0 + 32 * 4;
    const int voff_r0 = 4 * hi * RROW + (dir * 5 * HP + (2 * v) * 32 + li) * 4, voff_r1 = voff_r0 + 32 * 4;
    // LDS: this lane's row of the h tile for the fragment reads, its (row group, unit) for the cells' writes
    const int hx_off = (trow + li) * HS + 4 * hi;
    const int hy_off0 = (trow + 4 * hi) * HS + (2 * v) * 32 + li, hy_off1 = hy_off0 + 32;

    f32x16 acc0[4], acc1[4], c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;

    auto row0_of = [&](int step) { return ((size_t)(dir ? (T - 1 - step) : step)) * Bp + b0 + trow; };
    {   // gate pre-activations of step 0, both unit sets
        const rsrc_t rx = make_rsrc(a.xproj + row0_of(0) * (2 * GP), live_rows * XROW);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowc = (r & 3) + 8 * (r >> 2);
                acc0[g][r] = buf_load_x(rx, voff_x0, rowc * XROW + g * 128);
                acc1[g][r] = buf_load_x(rx, voff_x1, rowc * XROW + g * 128);
            }
    }
    __syncthreads();
    auto out_rsrc = [&](int step, rsrc_t& rh, rsrc_t& rr) {
        const size_t r = row0_of(step);
        rh = make_rsrc(a.hout + r * (2 * HP), live_rows * HROW);
        rr = SAVE ? make_rsrc(a.resv + r * (2 * 5 * HP), live_rows * RROW) : rh;
    };
    auto in_rsrc = [&](int step, bool on) {
        return make_rsrc(a.xproj + row0_of(on ? step : 0) * (2 * GP), on ? live_rows * XROW : 0);
    };
    rsrc_t rh, rr;
    // prologue: h_{-1} = 0, the step-0 pre-activations are final.  cell U0 (step 0) -> buffer 0, loads x_1[U0]
    out_rsrc(0, rh, rr);
    q_block<0, 3, false, SAVE>(acc1, hbuf + hx_off, wb0, lane, diag_hot, acc0, c0, hbuf + hy_off0, rh, rr, in_rsrc(1, T > 1), voff_h0, voff_r0,
                               voff_x0);
    AVSI_LDS_BARRIER();
    for (int step = 1; step < T; ++step) {
        float* hcur = hbuf + ((step - 1) & 1) * HBUF;
        float* hnext = hbuf + (step & 1) * HBUF;
        // Q00 || cell U1 (step - 1) -> hcur[U1], loads x_step[U1]
        out_rsrc(step - 1, rh, rr);
        q_block<0, 3, true, SAVE>(acc0, hcur + hx_off, wb0, lane, diag_hot, acc1, c1, hcur + hy_off1, rh, rr, in_rsrc(step, true), voff_h1,
                                  voff_r1, voff_x1);
        AVSI_LDS_BARRIER();
        // Q01
        q_block<1, 0, true, SAVE>(acc0, hcur + hx_off, wb0, lane, diag_hot, acc1, c1, hcur + hy_off1, rh, rr, rh, voff_h1, voff_r1, voff_x1);
        // Q11 || cell U0 (step) -> hnext[U0]
        out_rsrc(step, rh, rr);
        q_block<1, 1, true, SAVE>(acc1, hcur + hx_off, wb1, lane, diag_hot, acc0, c0, hnext + hy_off0, rh, rr, rh, voff_h0, voff_r0, voff_x0);
        // Q10 || loads x_{step+1}[U0]
        q_block<0, 2, true, SAVE>(acc1, hcur + hx_off, wb1, lane, diag_hot, acc0, c0, hnext + hy_off0, rh, rr, in_rsrc(step + 1, step + 1 < T),
                                  voff_h0, voff_r0, voff_x0);
        AVSI_LDS_BARRIER();
    }
    // epilogue: cell U1 (step T - 1)
    out_rsrc(T - 1, rh, rr);
    q_block<0, 1, false, SAVE>(acc0, hbuf + hx_off, wb0, lane, diag_hot, acc1, c1, hbuf + ((T - 1) & 1) * HBUF + hy_off1, rh, rr, rh, voff_h1,
                               voff_r1, voff_x1);
}

template <bool SAVE>
int launch_rec_q(const RecArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 64 * HS * 4;
    (void)hipFuncSetAttribute((const void*)blstm_rec_fwd_q_kernel<SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = (int)avsi_ceil_div(a.rend - a.row0, 64);
    hipLaunchKernelGGL((blstm_rec_fwd_q_kernel<SAVE>), dim3(tiles, 2), dim3(512), lds, st, a);
    return avsi_launch_status();
}

template <bool SAVE>
int launch_rec_pp(const RecArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 32 * HS * 4;
    const int tiles = (int)avsi_ceil_div(a.rend - a.row0, 64);
    // AVSI_PP_RESIDENT0=0 (A/B only): every phase requests its first fragment group again, as before round 6
    static const bool resident = !(getenv("AVSI_PP_RESIDENT0") && atoi(getenv("AVSI_PP_RESIDENT0")) == 0);
    if (resident) {
        (void)hipFuncSetAttribute((const void*)blstm_rec_fwd_pp_kernel<SAVE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((blstm_rec_fwd_pp_kernel<SAVE, true>), dim3(tiles, 2), dim3(512), lds, st, a);
    } else {
        (void)hipFuncSetAttribute((const void*)blstm_rec_fwd_pp_kernel<SAVE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((blstm_rec_fwd_pp_kernel<SAVE, false>), dim3(tiles, 2), dim3(512), lds, st, a);
    }
    return avsi_launch_status();
}

template <int MT, bool SAVE>
int launch_rec(const RecArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * MT * 32 * HS * 4;
    (void)hipFuncSetAttribute((const void*)blstm_rec_fwd_kernel<MT, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    const int tiles = (int)avsi_ceil_div(a.rend - a.row0, MT * 32);
    hipLaunchKernelGGL((blstm_rec_fwd_kernel<MT, SAVE>), dim3(tiles, 2), dim3(512), lds, st, a);
    return avsi_launch_status();
}

}  // namespace

extern "C" int avsi_blstm_rec_fwd_rows_f32(const float* xproj, const float* whp, float* hout, float* reserve, int T, int Bp,
                                           int rows_per_wg, int first_row, int rows, void* stream) {
    if (!xproj || !whp || !hout || T <= 0 || Bp <= 0) return AVSI_ERR_INVALID_ARG;
    if (Bp % 32) return AVSI_ERR_INVALID_ARG;  // batch is padded to whole 32-row MFMA tiles
    if (first_row < 0 || rows <= 0 || (first_row & 31) || (rows & 31) || first_row + rows > Bp) return AVSI_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(whp) & 15)) return AVSI_ERR_UNSUPPORTED;
    static const int q_diag = getenv("AVSI_REC_Q_DIAG") ? atoi(getenv("AVSI_REC_Q_DIAG")) : 0;
    RecArgs a{xproj, whp, hout, reserve, T, Bp, nullptr, first_row, first_row + rows, q_diag};
    // 64 rows per workgroup halves the Wh stream per flop; 32 rows spreads a small batch wider: up to 4096 utterances
    // the 32-row workgroups (two directions x rows / 32) fit the chip in one round, beyond that they would need a second
    // round where the 64-row kernel still needs one (6144 utterances, whole inference step: 131 -> 119 ms)
    int mt = rows_per_wg;
    if (mt == 0) mt = (rows > 32 * AVSI_NUM_CU / 2 && rows % 64 == 0) ? 64 : 32;
    if (mt != 32 && mt != 64 && mt != 65 && mt != 66) return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    const hipStream_t st = (hipStream_t)stream;
    if (mt == 64) return reserve ? launch_rec_pp<true>(a, st) : launch_rec_pp<false>(a, st);
    if (mt == 65) return reserve ? launch_rec<2, true>(a, st) : launch_rec<2, false>(a, st);  // A/B baseline
    if (mt == 66) return reserve ? launch_rec_q<true>(a, st) : launch_rec_q<false>(a, st);
    return reserve ? launch_rec<1, true>(a, st) : launch_rec<1, false>(a, st);
}

extern "C" int avsi_blstm_rec_fwd_f32(const float* xproj, const float* whp, float* hout, float* reserve, int T, int Bp,
                                      int rows_per_wg, void* stream) {
    return avsi_blstm_rec_fwd_rows_f32(xproj, whp, hout, reserve, T, Bp, rows_per_wg, 0, Bp, stream);
}

#ifdef AVSI_REC_STAMPS
// Diagnostic entry (tools/rec_stamps.cpp): same kernel with phase stamps written to `stamps`.
extern "C" int avsi_blstm_rec_fwd_stamps(const float* xproj, const float* whp, float* hout, int T, int Bp, int mt,
                                         unsigned long long* stamps, void* stream) {
    RecArgs a{xproj, whp, hout, nullptr, T, Bp, stamps, 0, Bp, 0};
    avsi_clear_error();
    return mt == 64 ? launch_rec<2, false>(a, (hipStream_t)stream) : launch_rec<1, false>(a, (hipStream_t)stream);
}
#endif
