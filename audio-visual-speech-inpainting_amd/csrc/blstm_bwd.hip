// Backward (BPTT) of the recurrent half of one bidirectional LSTM layer on gfx950.
//
// Gradient of the graph the reference differentiates with tf.gradients / the CudnnLSTM backward
// op (models.py:95-115, train_op at models.py:161-179).  For every step, walking each direction
// against its forward order:
//     dh   = dH_out[t] + dz_{next} . Wh^T                (next = the step processed just before)
//     do   = dh tanh(c);  dc = dh o (1 - tanh(c)^2) + dc_next
//     di   = dc j;  dj = dc i;  df = dc c_prev;  dc_next = dc f
//     dz   = (di i(1-i), dj (1-j^2), df f(1-f), do o(1-o))
// dz is written out in the packed gate-column order of xproj; the time-batched gradients
// (dX = dZ . Wx^T, dWx = X^T . dZ, dWh = H_prev^T . dZ, db = colsum dZ) are large GEMMs / column
// sums over dZ afterwards (gemm.hip, elementwise.hip).
//
// Mapping: same batch-stationary scheme as the forward kernel (a workgroup owns 32 utterances of
// one direction, wave w owns hidden units [32w, 32w+32) x 4 gates, everything elementwise is
// lane-local).  The recurrent product dz . Wh^T reduces over the 1024 gate columns, which are
// spread over all 8 waves, so dz crosses LDS once per step: [32][1024 + 4] floats (131.6 KB),
// written in C/D layout, read back as MFMA A fragments with ds_read_b128.  Wh^T (1 MiB per
// direction) streams from L2 in host-packed fragment order, SGPR base + lane addressing.
// Inputs of the NEXT step (dH_out, the five reserve planes, c_prev) are fetched into registers
// before the MFMA phase so their latency hides under it.
//
// Layouts (floats):
//   dhout [T][Bp][512]         gradient w.r.t. the layer output (fw 0..255, bw 256..511)
//   resv  [T][Bp][2][5][256]   i, j, f, o (activated), c_t from the forward pass
//   whbT  [2][8 w][128 q][64 lane][4 s] = Wh[unit' = 32w + (lane&31)][packed col = 8q + 4(lane>>5) + s]
//   dz    [T][Bp][2][1024]     packed gate columns (col = 128 w + 32 gate + u)
#include <stdlib.h>

#include "avsi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int HP = 256, GP = 4 * HP, NWAVE = 8;
constexpr int ZS = GP + 4;  // LDS row stride of the dz tile (conflict-free b128 reads)

struct BwdArgs {
    const float* dhout;
    const float* resv;
    const float* whbT;
    float* dz;
    int T, Bp;
};

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f* gptr4;
__device__ __forceinline__ gptr4 opaque_base(const float4* p) {
    gptr4 g = (gptr4)(const void*)p;
    asm volatile("" : "+s"(g));
    return g;
}
__device__ __forceinline__ float4 ldg4(gptr4 p, int idx) {
    const v4f v = p[idx];
    return make_float4(v.x, v.y, v.z, v.w);
}

struct StepIn {  // one step's inputs of a lane: 16 row-registers x (dH, i, j, f, o, c, c_prev)
    float dh[16], gi[16], gj[16], gf[16], go[16], c[16], cp[16];
};

__global__ __launch_bounds__(512, 2) void blstm_rec_bwd_kernel(const BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* zbuf = reinterpret_cast<float*>(smem);  // [32][ZS]
    constexpr int HROW = 2 * HP * 4, RROW = 2 * 5 * HP * 4, ZROW = 2 * GP * 4;  // row pitches, bytes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * 32;
    const int T = a.T, Bp = a.Bp;
    const int live_rows = min(32, Bp - b0);

    const float4* __restrict__ wb = reinterpret_cast<const float4*>(a.whbT) + (size_t)(dir * NWAVE + w) * (128 * 64);
    const int voff_h = 4 * hi * HROW + (dir * HP + w * 32 + li) * 4;
    const int voff_r = 4 * hi * RROW + (dir * 5 * HP + w * 32 + li) * 4;
    const int voff_z = 4 * hi * ZROW + (dir * GP + w * 128 + li) * 4;

    // backward step s visits forward time t: fw walks T-1 .. 0, bw walks 0 .. T-1
    auto time_of = [&](int s) { return dir ? s : (T - 1 - s); };

    auto load_step = [&](StepIn& in, int s) {
        const int t = time_of(s);
        const int tp = dir ? t + 1 : t - 1;                 // forward-previous step (owner of c_{prev})
        const bool has_prev = dir ? (t + 1 < T) : (t > 0);
        const size_t row0 = (size_t)t * Bp + b0;
        const rsrc_t rh = make_rsrc(a.dhout + row0 * (2 * HP), live_rows * HROW);
        const rsrc_t rr = make_rsrc(a.resv + row0 * (2 * 5 * HP), live_rows * RROW);
        const rsrc_t rp = make_rsrc(a.resv + ((size_t)(has_prev ? tp : t) * Bp + b0) * (2 * 5 * HP),
                                    has_prev ? live_rows * RROW : 0);   // zero records => c_prev = 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowc = (r & 3) + 8 * (r >> 2);
            in.dh[r] = buf_load(rh, voff_h, rowc * HROW);
            in.gi[r] = buf_load(rr, voff_r, rowc * RROW + 0 * HP * 4);
            in.gj[r] = buf_load(rr, voff_r, rowc * RROW + 1 * HP * 4);
            in.gf[r] = buf_load(rr, voff_r, rowc * RROW + 2 * HP * 4);
            in.go[r] = buf_load(rr, voff_r, rowc * RROW + 3 * HP * 4);
            in.c[r] = buf_load(rr, voff_r, rowc * RROW + 4 * HP * 4);
            in.cp[r] = buf_load(rp, voff_r, rowc * RROW + 4 * HP * 4);
        }
    };

    f32x16 dhrec;   // dz_{next} . Wh^T for this wave's 32 units (C/D layout)
    float dcn[16];  // dc_next
#pragma unroll
    for (int r = 0; r < 16; ++r) dhrec[r] = 0.f, dcn[r] = 0.f;

    StepIn cur;
    load_step(cur, 0);

    for (int s = 0; s < T; ++s) {
        const int t = time_of(s);
        const rsrc_t rz = make_rsrc(a.dz + ((size_t)t * Bp + b0) * (2 * GP), live_rows * ZROW);
        // ---- elementwise BPTT, lane-local
        float* zl = zbuf + (4 * hi) * ZS + w * 128 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowl = (r & 3) + 8 * (r >> 2);
            const float dh = cur.dh[r] + dhrec[r];
            const float ig = cur.gi[r], jg = cur.gj[r], fg = cur.gf[r], og = cur.go[r];
            const float tc = 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * cur.c[r])) - 1.f;
            const float dc = dh * og * (1.f - tc * tc) + dcn[r];
            dcn[r] = dc * fg;
            const float dzi = dc * jg * ig * (1.f - ig);
            const float dzj = dc * ig * (1.f - jg * jg);
            const float dzf = dc * cur.cp[r] * fg * (1.f - fg);
            const float dzo = dh * tc * og * (1.f - og);
            zl[rowl * ZS + 0] = dzi, zl[rowl * ZS + 32] = dzj, zl[rowl * ZS + 64] = dzf, zl[rowl * ZS + 96] = dzo;
            buf_store(rz, voff_z, rowl * ZROW + 0 * 128, dzi);
            buf_store(rz, voff_z, rowl * ZROW + 1 * 128, dzj);
            buf_store(rz, voff_z, rowl * ZROW + 2 * 128, dzf);
            buf_store(rz, voff_z, rowl * ZROW + 3 * 128, dzo);
        }
        AVSI_LDS_BARRIER();  // dz visible in LDS; its global stores keep draining under the MFMA phase
        if (s + 1 == T) break;
        // ---- next step's inputs: in flight during the MFMA phase
        load_step(cur, s + 1);
        // ---- dhrec = dz . Wh^T over the 1024 packed gate columns (128 k-groups of 8)
#pragma unroll
        for (int r = 0; r < 16; ++r) dhrec[r] = 0.f;
        float4 bw0, bw1, af0, af1;
        {
            const gptr4 wq = opaque_base(wb);
            bw0 = ldg4(wq, lane);
            af0 = *reinterpret_cast<const float4*>(zbuf + li * ZS + 4 * hi);
        }
        for (int q = 0; q < 128; q += 2) {
            {
                const gptr4 wq = opaque_base(wb + (q + 1) * 64);
                bw1 = ldg4(wq, lane);
                af1 = *reinterpret_cast<const float4*>(zbuf + li * ZS + 8 * (q + 1) + 4 * hi);
            }
            __builtin_amdgcn_sched_barrier(0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.x, bw0.x, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.y, bw0.y, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.z, bw0.z, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.w, bw0.w, dhrec, 0, 0, 0);
            if (q + 2 < 128) {
                const gptr4 wq = opaque_base(wb + (q + 2) * 64);
                bw0 = ldg4(wq, lane);
                af0 = *reinterpret_cast<const float4*>(zbuf + li * ZS + 8 * (q + 2) + 4 * hi);
            }
            __builtin_amdgcn_sched_barrier(0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.x, bw1.x, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.y, bw1.y, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.z, bw1.z, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.w, bw1.w, dhrec, 0, 0, 0);
        }
        AVSI_LDS_BARRIER();  // every wave is done reading zbuf before the next step overwrites it
    }
}

// ------------------------------------------------------------------------------------------------------------
// K-halved variant, built for TWO workgroups per CU.
// The kernel above keeps the whole [32][1024] dz tile in LDS (131.6 KB: one workgroup per CU), so while its eight
// waves run the elementwise phase (and wait for that phase's inputs) the matrix pipe idles: 60 % of fp32-MFMA peak.
// Here a workgroup passes dz through LDS in two halves of 512 columns -- gates (i, j), then (f, o); the second
// half waits in 32 registers per lane -- so its LDS tile is 66 KB and a second workgroup fits on the CU.  The two run
// out of phase by themselves: one's elementwise phase, input loads and barriers hide under the other's MFMAs.
// The reduction over the 1024 packed gate columns is only re-ordered: half h takes the 8-column groups q with
// (8 q mod 128) / 64 == h, i.e. for every wave's 128 columns first its i / j gates, then f / o -- the same Wh^T
// fragments (whbT is not repacked), another summation order.
// Registers: <= 128 (four waves per SIMD).  The step inputs (7 x 16 values per lane) do not fit beside the held
// half: they come in four chunks of four row-registers, the first requested during the second MFMA phase (the held
// half is gone by then), each of the others while the chunk before it is computed.
// ------------------------------------------------------------------------------------------------------------
constexpr int ZH = GP / 2 + 4;  // LDS row stride of a half tile

__global__ __launch_bounds__(512, 4) void blstm_rec_bwd_kh_kernel(const BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* zbuf = reinterpret_cast<float*>(smem);  // [32][ZH]
    constexpr int HROW = 2 * HP * 4, RROW = 2 * 5 * HP * 4, ZROW = 2 * GP * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * 32;
    const int T = a.T, Bp = a.Bp;
    const int live_rows = min(32, Bp - b0);

    const float4* __restrict__ wb = reinterpret_cast<const float4*>(a.whbT) + (size_t)(dir * NWAVE + w) * (128 * 64);
    const int voff_h = 4 * hi * HROW + (dir * HP + w * 32 + li) * 4;
    const int voff_r = 4 * hi * RROW + (dir * 5 * HP + w * 32 + li) * 4;
    const int voff_z = 4 * hi * ZROW + (dir * GP + w * 128 + li) * 4;
    auto time_of = [&](int s) { return dir ? s : (T - 1 - s); };

    constexpr int CH = 4;   // row-registers per chunk of step inputs
    struct RowsC {
        float dh[CH], gi[CH], gj[CH], gf[CH], go[CH], c[CH], cp[CH];
    };
    auto loadc = [&](RowsC& in, int s, int r0) {
        const int t = time_of(s);
        const int tp = dir ? t + 1 : t - 1;
        const bool has_prev = dir ? (t + 1 < T) : (t > 0);
        const size_t row0 = (size_t)t * Bp + b0;
        const rsrc_t rh = make_rsrc(a.dhout + row0 * (2 * HP), live_rows * HROW);
        const rsrc_t rr = make_rsrc(a.resv + row0 * (2 * 5 * HP), live_rows * RROW);
        const rsrc_t rp = make_rsrc(a.resv + ((size_t)(has_prev ? tp : t) * Bp + b0) * (2 * 5 * HP),
                                    has_prev ? live_rows * RROW : 0);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int r = r0 + e;
            const int rowc = (r & 3) + 8 * (r >> 2);
            in.dh[e] = buf_load(rh, voff_h, rowc * HROW);
            in.gi[e] = buf_load(rr, voff_r, rowc * RROW + 0 * HP * 4);
            in.gj[e] = buf_load(rr, voff_r, rowc * RROW + 1 * HP * 4);
            in.gf[e] = buf_load(rr, voff_r, rowc * RROW + 2 * HP * 4);
            in.go[e] = buf_load(rr, voff_r, rowc * RROW + 3 * HP * 4);
            in.c[e] = buf_load(rr, voff_r, rowc * RROW + 4 * HP * 4);
            in.cp[e] = buf_load(rp, voff_r, rowc * RROW + 4 * HP * 4);
        }
    };

    f32x16 dhrec;
    float dcn[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dhrec[r] = 0.f, dcn[r] = 0.f;
    float hold[32];  // dz of gates f, o: the second half, written to LDS once the first has been consumed

    // one half of the reduction: 64 groups of 8 columns, q = 16 (g / 8) + 8 h + g % 8
    auto mfma_half = [&](int h) {
        float4 bw0, bw1, af0, af1;
        auto qof = [&](int g) { return 16 * (g >> 3) + 8 * h + (g & 7); };
        {
            const gptr4 wq = opaque_base(wb + qof(0) * 64);
            bw0 = ldg4(wq, lane);
            af0 = *reinterpret_cast<const float4*>(zbuf + li * ZH + 4 * hi);
        }
#pragma unroll 4
        for (int g = 0; g < 64; g += 2) {
            {
                const gptr4 wq = opaque_base(wb + qof(g + 1) * 64);
                bw1 = ldg4(wq, lane);
                af1 = *reinterpret_cast<const float4*>(zbuf + li * ZH + 8 * (g + 1) + 4 * hi);
            }
            __builtin_amdgcn_sched_barrier(0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.x, bw0.x, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.y, bw0.y, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.z, bw0.z, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.w, bw0.w, dhrec, 0, 0, 0);
            if (g + 2 < 64) {
                const gptr4 wq = opaque_base(wb + qof(g + 2) * 64);
                bw0 = ldg4(wq, lane);
                af0 = *reinterpret_cast<const float4*>(zbuf + li * ZH + 8 * (g + 2) + 4 * hi);
            }
            __builtin_amdgcn_sched_barrier(0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.x, bw1.x, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.y, bw1.y, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.z, bw1.z, dhrec, 0, 0, 0);
            dhrec = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.w, bw1.w, dhrec, 0, 0, 0);
        }
    };

    // elementwise BPTT of one chunk: gates i, j go to the LDS half tile, f, o into `hold`; all four to HBM
    auto cellc = [&](const RowsC& in, int r0, const f32x16& rec, const rsrc_t rz) {
        float* zl = zbuf + (4 * hi) * ZH + w * 64 + li;
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int r = r0 + e;
            const int rowl = (r & 3) + 8 * (r >> 2);
            const float dh = in.dh[e] + rec[r];
            const float ig = in.gi[e], jg = in.gj[e], fg = in.gf[e], og = in.go[e];
            const float tc = 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * in.c[e])) - 1.f;
            const float dc = dh * og * (1.f - tc * tc) + dcn[r];
            dcn[r] = dc * fg;
            const float dzi = dc * jg * ig * (1.f - ig);
            const float dzj = dc * ig * (1.f - jg * jg);
            const float dzf = dc * in.cp[e] * fg * (1.f - fg);
            const float dzo = dh * tc * og * (1.f - og);
            zl[rowl * ZH + 0] = dzi, zl[rowl * ZH + 32] = dzj;
            hold[2 * r] = dzf, hold[2 * r + 1] = dzo;
            buf_store(rz, voff_z, rowl * ZROW + 0 * 128, dzi);
            buf_store(rz, voff_z, rowl * ZROW + 1 * 128, dzj);
            buf_store(rz, voff_z, rowl * ZROW + 2 * 128, dzf);
            buf_store(rz, voff_z, rowl * ZROW + 3 * 128, dzo);
        }
    };

    // (Delaying one of a CU's two workgroups by a fraction of a step at the start changes nothing: 19.5 ms for any
    // delay from 0 to 80 us.  Neither does keeping the two out of the elementwise phase at the same time -- a token per
    // physical CU (HW_REG_XCC_ID / HW_REG_HW_ID), taken with a scalar atomic at the top of the phase: 20.6 ms with and
    // without the token in a build whose extra control flow cost 15 spills, so the idle quarter of the matrix pipe is
    // not two coinciding elementwise phases.  PMC: the matrix pipe is busy 76 % of the kernel's cycles at the 2.19 GHz the chip holds
    // under this load; waves are parked on s_waitcnt / barriers 25 % of their time.)
    RowsC ca, cb;
    loadc(ca, 0, 0);
    for (int s = 0; s < T; ++s) {
        const int t = time_of(s);
        const rsrc_t rz = make_rsrc(a.dz + ((size_t)t * Bp + b0) * (2 * GP), live_rows * ZROW);
        // ---- elementwise phase, four chunks of four row-registers: a chunk is requested while the one before it
        //      is computed (the first came in during the second MFMA phase of the previous step; two chunks there
        //      push the MFMA loop over 128 registers: 28 spills)
        loadc(cb, s, 4);
        cellc(ca, 0, dhrec, rz);
        loadc(ca, s, 8);
        cellc(cb, 4, dhrec, rz);
        loadc(cb, s, 12);
        cellc(ca, 8, dhrec, rz);
        cellc(cb, 12, dhrec, rz);
        AVSI_LDS_BARRIER();        // half 0 (gates i, j) visible
        if (s + 1 == T) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) dhrec[r] = 0.f;
        mfma_half(0);
        AVSI_LDS_BARRIER();        // every wave is done reading half 0
        {
            float* zl = zbuf + (4 * hi) * ZH + w * 64 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowl = (r & 3) + 8 * (r >> 2);
                zl[rowl * ZH + 0] = hold[2 * r], zl[rowl * ZH + 32] = hold[2 * r + 1];
            }
        }
        AVSI_LDS_BARRIER();        // half 1 (gates f, o) visible
        loadc(ca, s + 1, 0);       // next step's first chunk: in flight during the second MFMA phase
        mfma_half(1);
        AVSI_LDS_BARRIER();        // done reading half 1 before the next step overwrites the tile
    }
}

}  // namespace

// blstm_bwd_pp.hip
int avsi_blstm_rec_bwd_pp_launch(const float* dhout, const float* reserve, const float* whbT, float* dz, int T, int Bp,
                                 hipStream_t st);

// Which batch-stationary BPTT kernel runs a batch of Bp utterances (AVSI_BWD_PP = 1 / 0 forces / forbids the ping-pong
// kernel).  The ping-pong kernel owns 64 utterances per workgroup and one workgroup per CU; the K-halved kernel 32
// utterances per workgroup and two per CU.  Measured per layer, T = 250, ms (ping-pong / K-halved): 6144 utterances
// 16.8 / 19.4, 8192: 18.5 / 19.7 (115.9 / 109.1 TFLOP/s), 12288: 36.4 / 29.8 -- the ping-pong kernel wins where its
// Bp / 64 x 2 workgroups fit the chip in ONE round and the K-halved kernel's Bp / 32 x 2 no longer fit half of its 512
// slots; beyond 8192 its second round costs more than it gains.
static bool bwd_use_pp(int Bp) {
    const char* e = getenv("AVSI_BWD_PP");       // read at every call: the tests switch it
    if (e && *e) return atoi(e) != 0;
    return Bp > 32 * AVSI_NUM_CU / 2 && Bp <= 64 * AVSI_NUM_CU / 2;       // 4096 < Bp <= 8192
}

// K-halved kernel (two workgroups per CU) by default; AVSI_BWD_KH=0 selects the whole-tile kernel (A/B runs)
static bool bwd_whole_tile() {
    static const bool whole_tile = getenv("AVSI_BWD_KH") && atoi(getenv("AVSI_BWD_KH")) == 0;
    return whole_tile;
}

extern "C" const char* avsi_blstm_rec_bwd_kernel_name(int Bp) {
    if (bwd_whole_tile()) return "blstm_rec_bwd_kernel";
    return bwd_use_pp(Bp) ? "blstm_rec_bwd_pp_kernel" : "blstm_rec_bwd_kh_kernel";
}

extern "C" int avsi_blstm_rec_bwd_f32(const float* dhout, const float* reserve, const float* whbT, float* dz, int T,
                                      int Bp, void* stream) {
    if (!dhout || !reserve || !whbT || !dz || T <= 0 || Bp <= 0) return AVSI_ERR_INVALID_ARG;
    if (Bp % 32) return AVSI_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(whbT) & 15) return AVSI_ERR_UNSUPPORTED;
    BwdArgs a{dhout, reserve, whbT, dz, T, Bp};
    avsi_clear_error();
    const bool whole_tile = bwd_whole_tile();
    if (!whole_tile && bwd_use_pp(Bp)) return avsi_blstm_rec_bwd_pp_launch(dhout, reserve, whbT, dz, T, Bp, (hipStream_t)stream);
    if (!whole_tile) {
        const size_t lds = (size_t)32 * ZH * 4;
        (void)hipFuncSetAttribute((const void*)blstm_rec_bwd_kh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(blstm_rec_bwd_kh_kernel, dim3(Bp / 32, 2), dim3(512), lds, (hipStream_t)stream, a);
        return avsi_launch_status();
    }
    const size_t lds = (size_t)32 * ZS * 4;
    (void)hipFuncSetAttribute((const void*)blstm_rec_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(blstm_rec_bwd_kernel, dim3(Bp / 32, 2), dim3(512), lds, (hipStream_t)stream, a);
    return avsi_launch_status();
}
