// BPTT of the recurrent half of one bidirectional LSTM layer, ping-pong form for large batches (gfx950).
//
// Same arithmetic and layouts as blstm_bwd.hip (gradient of the graph of reference models.py:95-115, train_op at
// models.py:161-179); what changes is WHO does what WHEN.  A workgroup owns 64 utterances of one direction as two 32-row
// tiles that alternate roles, the shape of the forward blstm_rec_fwd_pp_kernel:
//     phase A(s): MFMA tile 0: dhrec_0 <- dz_0(s) . Wh^T    ||  cell tile 1, step s     (uses dhrec_1 of phase B(s - 1))
//     phase B(s): MFMA tile 1: dhrec_1 <- dz_1(s) . Wh^T    ||  cell tile 0, step s + 1 (uses dhrec_0 of phase A(s))
// The elementwise BPTT of one tile, the loads of its seven inputs per cell and the stores of its dz sit INSIDE the other
// tile's 512-MFMA stream, so nothing but two barriers per phase (and the 64 LDS writes per lane between them) happens with
// the matrix pipe idle.  The whole [32][1024] dz tile of the MFMA tile lives in LDS (131.6 KB: one workgroup per CU, 256
// registers per lane); the cell tile's dz waits in 64 registers until the phase ends.
//
// The K-halved kernel of blstm_bwd.hip (two workgroups per CU, 128 registers) reached 0.62 - 0.66 of the fp32-MFMA peak
// with the matrix pipe busy 76 % of the cycles: its waves were parked on s_waitcnt a quarter of their time.  vmcnt counts
// loads AND stores in order, and its Wh^T fragments were requested ONE group of 4 MFMAs ahead (the BPTT product has one
// accumulator tile per wave where the forward product has four gate tiles, so a group is a quarter as long): the wait for
// a fragment requested just after a chunk of cell inputs was a wait for those HBM loads.  Here the fragments come through a
// ring of sixteen registers, FOURTEEN groups (56 MFMAs, ~3 us) ahead: the wait for a fragment allows everything younger
// than it -- cell inputs, dz stores -- to stay in flight, and the cell inputs requested just before a fragment have
// fourteen groups to arrive from HBM before that fragment is waited for (with six groups ahead the kernel spent 20 % of
// its time in exactly those waits; with the cell's loads switched off it ran at 138 TFLOP/s).
#include <stdio.h>
#include <stdlib.h>

#include "avsi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int HP = 256, GP = 4 * HP, NWAVE = 8;
constexpr int ZS = GP + 4;     // LDS row stride of the dz tile (conflict-free b128 reads)
constexpr int HROW = 2 * HP * 4, RROW = 2 * 5 * HP * 4, ZROW = 2 * GP * 4;      // row pitches, bytes
constexpr int RING = 8;        // Wh^T fragment registers (float4 each)
constexpr int AHEAD = 6;       // Wh^T fragments in flight, in groups of 4 MFMAs (14 of a ring of 16 measured the same)

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
// AUX = 2: the `nt` (non-temporal, streaming) cache policy.  Everything this kernel reads or writes through these two is
// touched ONCE (46 GB per launch at 8192 utterances), while the 2 MB of Wh^T are re-read by every workgroup every phase
// and must stay in the XCD's 4 MB L2.
#ifndef AVSI_PP_AUX
#define AVSI_PP_AUX 2
#endif
__device__ __forceinline__ float buf_load(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AVSI_PP_AUX));
}
__device__ __forceinline__ void buf_store(rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, AVSI_PP_AUX);
}
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f* gptr4;
__device__ __forceinline__ gptr4 opaque_base(const float4* p) {
    gptr4 g = (gptr4)(const void*)p;
    asm volatile("" : "+s"(g));
    return g;
}
// (a running pointer, opaque after every advance: 128 bases `wb + g * 64` of a phase are loop-invariant, the compiler computed
// them all ahead of the time loop, parked the 256 SGPR values in VGPR lanes and fetched each pair back with two v_readlane
// per group of 4 MFMAs -- 489 of the kernel's 2900 vector instructions per pair of phases; s_add_u32 / s_addc_u32 instead)
__device__ __forceinline__ gptr4 opaque_next(gptr4 g, int float4s) {
    g += float4s;
    asm volatile("" : "+s"(g));
    return g;
}
__device__ __forceinline__ float4 ldg4(gptr4 p, int idx) {
    const v4f v = p[idx];
    return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int CH = 2;      // row-registers per chunk of cell inputs (eight chunks per cell)
struct RowsC {
    float dh[CH], gi[CH], gj[CH], gf[CH], go[CH], cp[CH];      // (c_t itself is last step's c_prev: carried in registers)
};

// Where a (tile, step) lives.  The buffer descriptors are built from these few scalars AT EVERY USE (a dozen scalar
// instructions) instead of being carried through a phase: eight live descriptors are 32 SGPRs, the phase ran out of them,
// the compiler parked descriptors in VGPRs and wrapped every buffer access in a waterfall loop (256 of them: the cell
// work then cost 30 % of the kernel where its arithmetic is worth 3).
struct Ctx {
    const float* dhout;
    const float* resv;
    float* dz;
    int T, Bp, b0, dir;
};
struct Step {      // backward step s of a tile; on = false: descriptors of zero records (loads return 0, stores are dropped)
    int tile, s, on;
};
// (Bp is a multiple of 32: a tile has 32 live rows or none.  Integer arithmetic on purpose: min(max()) of uniform
// values is matched to v_med3_i32 and `cond ? rows : 0` to v_cndmask -- VALU results, a descriptor in VGPRs, and every
// buffer access in a waterfall loop.)
__device__ __forceinline__ int live_rows(const Ctx& c, const Step& w) {
    const int in_batch = (int)((unsigned)(c.b0 + 32 * w.tile) < (unsigned)c.Bp);
    const int in_time = (int)((unsigned)w.s < (unsigned)c.T);
    return 32 * (in_batch & in_time & w.on);
}
// backward step s visits forward time t: fw walks T-1 .. 0, bw walks 0 .. T-1
__device__ __forceinline__ int time_of(const Ctx& c, const Step& w) {
    const int sc = w.s * (int)((unsigned)w.s < (unsigned)c.T);      // out-of-range steps (zero-record descriptors): any valid row
    return sc + (1 - c.dir) * (c.T - 1 - 2 * sc);
}
__device__ __forceinline__ rsrc_t rsrc_h(const Ctx& c, const Step& w) {
    const size_t row0 = (size_t)time_of(c, w) * c.Bp + c.b0 + 32 * w.tile;
    return make_rsrc(c.dhout + row0 * (2 * HP), live_rows(c, w) * HROW);
}
__device__ __forceinline__ rsrc_t rsrc_r(const Ctx& c, const Step& w) {
    const size_t row0 = (size_t)time_of(c, w) * c.Bp + c.b0 + 32 * w.tile;
    return make_rsrc(c.resv + row0 * (2 * 5 * HP), live_rows(c, w) * RROW);
}
__device__ __forceinline__ rsrc_t rsrc_p(const Ctx& c, const Step& w) {      // the forward-previous step (owner of c_prev)
    const int t = time_of(c, w);
    const int tp = c.dir ? t + 1 : t - 1;
    const bool has_prev = c.dir ? (t + 1 < c.T) : (t > 0);
    return make_rsrc(c.resv + ((size_t)(t + (int)has_prev * (tp - t)) * c.Bp + c.b0 + 32 * w.tile) * (2 * 5 * HP),
                     (int)has_prev * live_rows(c, w) * RROW);
}
__device__ __forceinline__ rsrc_t rsrc_z(const Ctx& c, const Step& w) {
    const size_t row0 = (size_t)time_of(c, w) * c.Bp + c.b0 + 32 * w.tile;
    return make_rsrc(c.dz + row0 * (2 * GP), live_rows(c, w) * ZROW);
}

// One phase.  MFMA tile X (DO_MFMA): dhrec[X] <- zbuf . Wh^T over the 1024 packed gate columns, 128 groups of 8.
// Cell tile Y (DO_CELL): one row-register every 8 groups, its chunk of inputs (two row-registers) requested 20 groups
// before its first use; `nx` describes the tile whose cell runs in the NEXT phase: its first chunk is requested at group 112.
// DIAG (diagnostic builds of the kernel, AVSI_BWD_PP_DIAG; results are then WRONG): 1 = no cell arithmetic / dz stores,
// 2 = no cell input loads, 4 = no dz stores, 8 = Wh^T fragments loaded once per phase, 16 = no publish,
// 4096 = the fragment pointers as 128 loop-invariant bases (what the kernel did before: results right, v_readlane pairs).
template <int X, bool DO_MFMA, bool DO_CELL, int DIAG, bool CIRC>
__device__ __forceinline__ void bwd_pp_phase(f32x16 (&dhrec)[2], float (&dcn)[2][16], float (&ccar)[2][16], float (&dzh)[64],
                                             RowsC& ca, RowsC& cb,
                                             const float* __restrict__ zbuf, const float4* __restrict__ wb, float4 (&bw)[RING],
                                             const int lane, const int li, const int hi, const Ctx ctx, const Step cy, const Step nx,
                                             const int voff_h, const int voff_r, const int voff_z) {
    constexpr int Y = 1 - X;
    auto loadc = [&](RowsC& in, const Step& who0, int r0) {
        if (DIAG & 2) {          // no loads: opaque register values instead (the arithmetic stays)
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                float v = 0.25f + 0.001f * (r0 + e);
                asm volatile("v_mov_b32 %0, %0" : "+v"(v));
                in.dh[e] = v, in.gi[e] = v, in.gj[e] = v, in.gf[e] = v, in.go[e] = v, in.cp[e] = v;
            }
            return;
        }
        const Step who = (DIAG & 64) ? Step{who0.tile, who0.s & 1, who0.on} : who0;      // 64: always the same two steps (cache hits)
        struct {
            rsrc_t rh, rr, rp;
        } d{rsrc_h(ctx, who), rsrc_r(ctx, who), rsrc_p(ctx, who)};
        if (DIAG & 1024) {
            // timing experiment (results WRONG): the chunk's 12 values per lane by THREE 16-byte loads whose lanes run along
            // the unit axis (8 lanes per 128-byte row segment) -- the vector-memory side of a cell that is fed through LDS
            const size_t row0 = (size_t)time_of(ctx, who) * ctx.Bp + ctx.b0 + 32 * who.tile;
            const int l = (voff_r >> 2) & 31, hi4 = (voff_r >> 2) / (2 * 5 * HP), wq = ((voff_r >> 2) & 255) >> 5;
            const int lane64 = l + 8 * hi4;                      // 0 .. 63
            const int trow = ((r0 >> 1) * 8 + (lane64 >> 3)) & 31;      // inside the tile's own 32 rows
            const float* base = ctx.resv + (row0 + trow) * (2 * 5 * HP) + ctx.dir * 5 * HP + wq * 32 + (lane64 & 7) * 4;
            const v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(base));
            const v4f b = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(base + 2 * HP));
            const v4f c = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(base + 4 * HP));
            in.dh[0] = a.x, in.gi[0] = a.y, in.gj[0] = a.z, in.gf[0] = a.w, in.go[0] = b.x, in.cp[0] = b.y;
            in.dh[1] = b.z, in.gi[1] = b.w, in.gj[1] = c.x, in.gf[1] = c.y, in.go[1] = c.z, in.cp[1] = c.w;
            return;
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int r = r0 + e;
            const int rowc = (r & 3) + 8 * (r >> 2);
            in.dh[e] = buf_load(d.rh, voff_h, rowc * HROW);
            if (DIAG & 128) {       // one array of seven from memory
                float v = 0.25f + 0.001f * r;
                asm volatile("v_mov_b32 %0, %0" : "+v"(v));
                in.gi[e] = v, in.gj[e] = v, in.gf[e] = v, in.go[e] = v, in.cp[e] = v;
                continue;
            }
            if (DIAG & 512) {
                // timing experiment (results WRONG): the four gates by ONE 16-byte load at a 32-byte lane stride -- what a
                // reserve laid out [row][unit][i, j, f, o, c, pad x 3] would cost
                const int unit = ((voff_r >> 2) & 255) % 160, hi4 = (voff_r >> 2) / (2 * 5 * HP);
                const size_t row0 = (size_t)time_of(ctx, who) * ctx.Bp + ctx.b0 + 32 * who.tile;
                const v4f* gp = reinterpret_cast<const v4f*>(ctx.resv + (row0 + rowc + hi4) * (2 * 5 * HP) + ctx.dir * 5 * HP + unit * 8);
                const v4f g4 = __builtin_nontemporal_load(gp);
                in.gi[e] = g4.x, in.gj[e] = g4.y, in.gf[e] = g4.z, in.go[e] = g4.w;
            } else {
                in.gi[e] = buf_load(d.rr, voff_r, rowc * RROW + 0 * HP * 4);
                in.gj[e] = buf_load(d.rr, voff_r, rowc * RROW + 1 * HP * 4);
                in.gf[e] = buf_load(d.rr, voff_r, rowc * RROW + 2 * HP * 4);
                in.go[e] = buf_load(d.rr, voff_r, rowc * RROW + 3 * HP * 4);
            }
            if (DIAG & 256) {       // five of seven
                float v = 0.25f + 0.001f * r;
                asm volatile("v_mov_b32 %0, %0" : "+v"(v));
                in.cp[e] = v;
                continue;
            }
            in.cp[e] = buf_load(d.rp, voff_r, rowc * RROW + 4 * HP * 4);
        }
    };
    // The elementwise BPTT of one (row-register, lane) cell, cut into EIGHT stages, one per group of 4 MFMAs: the two waves
    // of a SIMD run the same instruction stream almost in step (they leave every barrier together), so a cell issued in
    // one piece -- ~150 instructions, ~700 cycles -- was 700 cycles in which NEITHER wave fed the matrix pipe, sixteen
    // times per phase (a fifth of the kernel).  A stage is 3 - 8 VALU instructions: it fits in the shadow of one MFMA.
    struct {
        float dh, x, rc, tc, dc, dzi, dzj, dzf, dzo;
    } st;
    // (pure arithmetic has no place of its own: the optimiser sinks it to its first use with a side effect -- the stores of
    // stages 6 / 7 -- and the stages collapse into one burst again.  `pin` makes a value opaque where its stage ends.)
    auto pin = [](float& v) { asm volatile("" : "+v"(v)); };
    auto cell_stage = [&](const RowsC& in, int r, int stage) {
        const int e = r & (CH - 1);
        const int rowl = (r & 3) + 8 * (r >> 2);
        if (DIAG & 1) {      // keep the products alive, nothing else
            if (stage == 0) dcn[Y][r] += dhrec[Y][r];
            return;
        }
        const float ig = in.gi[e], jg = in.gj[e], fg = in.gf[e], og = in.go[e];
        if (stage == 0) {
            st.dh = in.dh[e] + dhrec[Y][r];
            st.x = (DIAG & 32) ? ccar[Y][r] : __expf(-2.f * ccar[Y][r]);
            pin(st.dh), pin(st.x);
        } else if (stage == 1) {
            st.rc = (DIAG & 32) ? st.x : __builtin_amdgcn_rcpf(1.f + st.x);
            pin(st.rc);
        } else if (stage == 2) {
            st.tc = 2.f * st.rc - 1.f;
            st.dc = st.dh * og * (1.f - st.tc * st.tc) + dcn[Y][r];
            pin(st.tc), pin(st.dc);
        } else if (stage == 3) {
            dcn[Y][r] = st.dc * fg;
            st.dzo = st.dh * st.tc * og * (1.f - og);
            pin(dcn[Y][r]), pin(st.dzo);
        } else if (stage == 4) {
            st.dzi = st.dc * jg * ig * (1.f - ig);
            st.dzj = st.dc * ig * (1.f - jg * jg);
            pin(st.dzi), pin(st.dzj);
        } else if (stage == 5) {
            st.dzf = st.dc * in.cp[e] * fg * (1.f - fg);
            ccar[Y][r] = in.cp[e];          // the c_t of this tile's next backward step
            pin(st.dzf);
            dzh[4 * r + 0] = st.dzi, dzh[4 * r + 1] = st.dzj, dzh[4 * r + 2] = st.dzf, dzh[4 * r + 3] = st.dzo;
        } else if (DIAG & 2048) {
            // timing experiment (results WRONG): the cell's four dz values by one 16-byte store every FOURTH row-register,
            // lanes along the packed gate columns (a quarter of the store instructions, whole 128-byte segments)
            if (stage == 6 && (r & 3) == 3) {
                const size_t row0 = (size_t)time_of(ctx, cy) * ctx.Bp + ctx.b0 + 32 * cy.tile;
                const int l = (voff_z >> 2) & 31, hi4 = (voff_z >> 2) / (2 * GP), wq = ((voff_z >> 2) & 1023) >> 7;
                const int lane64 = l + 8 * hi4;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float* p = ctx.dz + (row0 + (r >> 2) * 8 + (lane64 >> 5) + 2 * u) * (2 * GP) + ctx.dir * GP + wq * 128 + (lane64 & 31) * 4;
                    __builtin_nontemporal_store((v4f){st.dzi, st.dzj, st.dzf, st.dzo}, reinterpret_cast<v4f*>(p));
                }
            }
        } else if (!(DIAG & 4)) {
            const rsrc_t rz = rsrc_z(ctx, cy);
            if (stage == 6) {
                buf_store(rz, voff_z, rowl * ZROW + 0 * 128, st.dzi);
                buf_store(rz, voff_z, rowl * ZROW + 1 * 128, st.dzj);
            } else {
                buf_store(rz, voff_z, rowl * ZROW + 2 * 128, st.dzf);
                buf_store(rz, voff_z, rowl * ZROW + 3 * 128, st.dzo);
            }
        }
    };

    // The fragment ring is CIRCULAR over the phases (round 6): every MFMA phase reads the same 128 groups, so the last AHEAD
    // requests of a phase are the first AHEAD groups of the next one, already in the ring (`bw` lives in the kernel) when that
    // phase begins behind its barrier -- a phase used to open with AHEAD loads and nothing to issue until they were back
    // from L2 (the forward kernel's version of this, its first group kept in registers, took 4 % off a launch).
    float4 af[2];
    gptr4 wrun = opaque_base(wb + (CIRC ? AHEAD * 64 : 0));   // the fragment group requested next
    if (DO_MFMA) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dhrec[X][r] = 0.f;
        if (!CIRC) {            // (AVSI_BWD_PP_CIRC=0, A/B only: the phase requests its first groups itself, as before round 6)
#pragma unroll
            for (int g = 0; g < AHEAD; ++g) {
                bw[g] = ldg4(wrun, lane);
                wrun = opaque_next(wrun, 64);
            }
        }
        af[0] = *reinterpret_cast<const float4*>(zbuf + li * ZS + 4 * hi);
    }
#pragma clang loop unroll(full)
    for (int qo = 0; qo < 16; ++qo)
#pragma clang loop unroll(full)
    for (int qi = 0; qi < 8; ++qi) {
        const int q = 8 * qo + qi;
        // ---- cell tile: requests first (they are the youngest entries of the in-order vmcnt queue: no wait of the MFMA
        //      stream below covers them), one row-register of arithmetic every 8 groups
        if (DO_CELL && (q & 15) == 0 && q < 112) {       // chunk q / 16 + 1, into the buffer chunk q / 16 - 1 left at group q - 3
            if (((q >> 4) & 1) == 0) loadc(cb, cy, CH * ((q >> 4) + 1));
            else loadc(ca, cy, CH * ((q >> 4) + 1));
        }
        if (q == 112) loadc(ca, nx, 0);          // first chunk of the next phase's cell (zero-record descriptors: none)
        if (DO_MFMA) {
            if ((CIRC || q + AHEAD < 128) && !((DIAG & 8) && q + AHEAD >= 8)) {
                if (q + AHEAD == 128) wrun = opaque_base(wb);           // around: the next phase's first groups
                if (DIAG & 4096) {
                    bw[(q + AHEAD) & (RING - 1)] = ldg4(opaque_base(wb + ((q + AHEAD) & 127) * 64), lane);
                } else {
                    bw[(q + AHEAD) & (RING - 1)] = ldg4(wrun, lane);
                    wrun = opaque_next(wrun, 64);
                }
            }
            if (q + 1 < 128) af[(q + 1) & 1] = *reinterpret_cast<const float4*>(zbuf + li * ZS + 8 * (q + 1) + 4 * hi);
            __builtin_amdgcn_sched_barrier(0);
            const float4 a4 = af[q & 1], b4 = bw[q & (RING - 1)];
            dhrec[X] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, dhrec[X], 0, 0, 0);
            dhrec[X] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, dhrec[X], 0, 0, 0);
            dhrec[X] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, dhrec[X], 0, 0, 0);
            dhrec[X] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, dhrec[X], 0, 0, 0);
        }
        if (DO_CELL) {                           // stage q % 8 of row-register r = q / 8
            const int r = q >> 3;
            if (((r / CH) & 1) == 0) cell_stage(ca, r, q & 7);
            else cell_stage(cb, r, q & 7);
        }
    }
}

template <int DIAG, bool CIRC = true>
__global__ __launch_bounds__(512, 2) void blstm_rec_bwd_pp_kernel(const BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* zbuf = reinterpret_cast<float*>(smem);  // [32][ZS]: dz of the tile whose MFMA phase comes next

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * 64;
    const int T = a.T, Bp = a.Bp;

    const float4* __restrict__ wb = reinterpret_cast<const float4*>(a.whbT) + (size_t)(dir * NWAVE + w) * (128 * 64);
    const int voff_h = 4 * hi * HROW + (dir * HP + w * 32 + li) * 4;
    const int voff_r = 4 * hi * RROW + (dir * 5 * HP + w * 32 + li) * 4;
    const int voff_z = 4 * hi * ZROW + (dir * GP + w * 128 + li) * 4;

    const Ctx ctx{a.dhout, a.resv, a.dz, T, Bp, b0, dir};
    // the cell tile's dz (C/D layout in registers) -> the LDS tile the next MFMA phase reads
    float dzh[64];
    auto publish = [&]() {
        if (DIAG & 16) return;
        float* zl = zbuf + (4 * hi) * ZS + w * 128 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowl = (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) zl[rowl * ZS + 32 * g] = dzh[4 * r + g];
        }
    };

    f32x16 dhrec[2];
    float dcn[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dhrec[m][r] = 0.f, dcn[m][r] = 0.f;

    float ccar[2][16];        // c_t of each tile's NEXT cell: loaded once for step 0, afterwards the c_prev of the step before
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const rsrc_t rr = rsrc_r(ctx, Step{m, 0, 1});
#pragma unroll
        for (int r = 0; r < 16; ++r) ccar[m][r] = buf_load(rr, voff_r, ((r & 3) + 8 * (r >> 2)) * RROW + 4 * HP * 4);
    }
    RowsC ca, cb;
    {   // first chunk of the prologue's cell (tile 0, step 0)
        const Step first{0, 0, 1};
        struct {
            rsrc_t rh, rr, rp;
        } d{rsrc_h(ctx, first), rsrc_r(ctx, first), rsrc_p(ctx, first)};
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int rowc = (e & 3) + 8 * (e >> 2);
            ca.dh[e] = buf_load(d.rh, voff_h, rowc * HROW);
            ca.gi[e] = buf_load(d.rr, voff_r, rowc * RROW + 0 * HP * 4);
            ca.gj[e] = buf_load(d.rr, voff_r, rowc * RROW + 1 * HP * 4);
            ca.gf[e] = buf_load(d.rr, voff_r, rowc * RROW + 2 * HP * 4);
            ca.go[e] = buf_load(d.rr, voff_r, rowc * RROW + 3 * HP * 4);
            ca.cp[e] = buf_load(d.rp, voff_r, rowc * RROW + 4 * HP * 4);
        }
    }
    float4 bw[RING];          // the Wh^T fragment ring, circular over the phases: its first AHEAD groups for the first MFMA phase
#pragma unroll
    for (int g = 0; g < RING; ++g) bw[g] = g < AHEAD ? wb[g * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    // prologue: cell tile 0, step 0 (dhrec = 0), no MFMA; then request the first chunk of tile 1, step 0
    bwd_pp_phase<1, false, true, DIAG, CIRC>(dhrec, dcn, ccar, dzh, ca, cb, zbuf, wb, bw, lane, li, hi, ctx, Step{0, 0, 1}, Step{1, 0, 1}, voff_h,
                                 voff_r, voff_z);
    publish();
    AVSI_LDS_BARRIER();
    for (int s = 0; s + 1 < T; ++s) {
        // phase A: MFMA tile 0 (dz_0(s) -> dhrec_0 for step s + 1) || cell tile 1, step s
        bwd_pp_phase<0, true, true, DIAG, CIRC>(dhrec, dcn, ccar, dzh, ca, cb, zbuf, wb, bw, lane, li, hi, ctx, Step{1, s, 1}, Step{0, s + 1, 1},
                                    voff_h, voff_r, voff_z);
        AVSI_LDS_BARRIER();       // every wave is done reading dz_0(s)
        publish();                // dz_1(s)
        AVSI_LDS_BARRIER();
        // phase B: MFMA tile 1 (dz_1(s) -> dhrec_1 for step s + 1) || cell tile 0, step s + 1
        bwd_pp_phase<1, true, true, DIAG, CIRC>(dhrec, dcn, ccar, dzh, ca, cb, zbuf, wb, bw, lane, li, hi, ctx, Step{0, s + 1, 1}, Step{1, s + 1, 1},
                                    voff_h, voff_r, voff_z);
        AVSI_LDS_BARRIER();
        publish();                // dz_0(s + 1)
        AVSI_LDS_BARRIER();
    }
    // epilogue: cell tile 1, step T - 1 (its dhrec comes from the last phase B; for T = 1 it is zero)
    bwd_pp_phase<0, false, true, DIAG, CIRC>(dhrec, dcn, ccar, dzh, ca, cb, zbuf, wb, bw, lane, li, hi, ctx, Step{1, T - 1, 1}, Step{0, T, 0}, voff_h,
                                 voff_r, voff_z);
    if (DIAG && T < 0) {        // never true: the diagnostic variants must not lose their arithmetic to dead-code elimination
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += dcn[0][r] + dcn[1][r] + dhrec[0][r] + dhrec[1][r];
        a.dz[tid] = s;
    }
}

}  // namespace

// internal (blstm_bwd.hip): the ping-pong BPTT kernel, 64 utterances per workgroup
int avsi_blstm_rec_bwd_pp_launch(const float* dhout, const float* reserve, const float* whbT, float* dz, int T, int Bp,
                                 hipStream_t st) {
    BwdArgs a{dhout, reserve, whbT, dz, T, Bp};
    const size_t lds = (size_t)32 * ZS * 4;
#define AVSI_PP_LAUNCH(D)                                                                                                  \
    do {                                                                                                                   \
        (void)hipFuncSetAttribute((const void*)blstm_rec_bwd_pp_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(blstm_rec_bwd_pp_kernel<D>, dim3((Bp + 63) / 64, 2), dim3(512), lds, st, a);                    \
    } while (0)
#ifdef AVSI_DIAG_KERNELS
    // `make EXTRA=-DAVSI_DIAG_KERNELS` only: AVSI_BWD_PP_DIAG selects variants with parts of the work taken out (HISTORY.md
    // 4.3g) -- their results are WRONG by design, so the production library does not contain them
    const int diag = getenv("AVSI_BWD_PP_DIAG") ? atoi(getenv("AVSI_BWD_PP_DIAG")) : 0;
    static bool warned = false;
    if (diag && !warned) {
        warned = true;
        fprintf(stderr, "avsi: AVSI_BWD_PP_DIAG=%d -- a diagnostic BPTT variant is active, its results are not valid\n", diag);
    }
    switch (diag) {
        case 4096: AVSI_PP_LAUNCH(4096); break;
        case 1: AVSI_PP_LAUNCH(1); break;
        case 3: AVSI_PP_LAUNCH(3); break;
        case 4: AVSI_PP_LAUNCH(4); break;
        case 8: AVSI_PP_LAUNCH(8); break;
        case 16: AVSI_PP_LAUNCH(16); break;
        case 31: AVSI_PP_LAUNCH(31); break;
        case 2: AVSI_PP_LAUNCH(2); break;
        case 64: AVSI_PP_LAUNCH(64); break;
        case 128: AVSI_PP_LAUNCH(128); break;
        case 512: AVSI_PP_LAUNCH(512); break;
        case 1024: AVSI_PP_LAUNCH(1024); break;
        case 2048: AVSI_PP_LAUNCH(2048); break;
        case 3072: AVSI_PP_LAUNCH(3072); break;
        case 516: AVSI_PP_LAUNCH(516); break;
        case 256: AVSI_PP_LAUNCH(256); break;
        case 68: AVSI_PP_LAUNCH(68); break;
        case 6: AVSI_PP_LAUNCH(6); break;
        case 32: AVSI_PP_LAUNCH(32); break;
        case 38: AVSI_PP_LAUNCH(38); break;
        case 22: AVSI_PP_LAUNCH(22); break;
        default: AVSI_PP_LAUNCH(0); break;
    }
#else
    // AVSI_BWD_PP_CIRC=0 (A/B only): every MFMA phase requests its first fragment groups itself, as before round 6
    static const bool circ = !(getenv("AVSI_BWD_PP_CIRC") && atoi(getenv("AVSI_BWD_PP_CIRC")) == 0);
    if (circ) {
        AVSI_PP_LAUNCH(0);
    } else {
        (void)hipFuncSetAttribute((const void*)blstm_rec_bwd_pp_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((blstm_rec_bwd_pp_kernel<0, false>), dim3((Bp + 63) / 64, 2), dim3(512), lds, st, a);
    }
#endif
#undef AVSI_PP_LAUNCH
    return avsi_launch_status();
}
