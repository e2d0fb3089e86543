// LWS sweeps with TWO utterances in the lanes of a wave -- the large-batch form of lws_skew.hip (avsi_lws_run_duo_f32; the
// phase refinement of the reference's `infer`, inference.py:119,141-154; lws.hip has the algorithm and its conventions,
// lws_skew.hip the skewed-frame pipeline this kernel keeps: rings, mirror rules, diagonal layout, counted waits, stages).
//
// lws_skew_kernel puts 64 consecutive frames of ONE utterance in the lanes of a wave, lane j six bins behind lane j - 1 (the
// dependence).  A lane then needs 286 steps for a frame but gets its next frame only every 64 x 6 = 384 steps: 45 of 64
// lanes are busy at any time, and at large batches the kernel is bound by the vector instructions it issues (with every row
// index masked into a window the L2 holds it takes 49 instead of 55 ms for 1024 utterances: HISTORY 4.3f), so a quarter of
// the chip's issue slots work on nothing.
//
// Here a wave carries two utterances in its two halves of 32 lanes, lane j of a half NINE bins behind lane j - 1: a lane gets
// its next frame every 32 x 9 = 288 steps and is busy for 279 of them.  What changes against lws_skew.hip:
//   * the old-value ring R looks 15 positions ahead instead of 11 (the row-below sum `up` that lane j makes for lane j - 1
//     belongs to a bin nine ahead of its own, and it is made one step early): rings of 15, a 15-step unrolled body, progress
//     published in fifths of a body (3 steps);
//   * the row-above sum `dn` that lane j makes for lane j + 1 is the one of position tau - 8 (all of it two steps old:
//     nothing of it waits for the bin of this step);
//   * both travel through LDS -- a rotate inside 32 lanes does not exist as a DPP -- written by every lane for its
//     neighbour's NEXT step and read back in the same step (one row of 64 complex values each per wave; LDS operations of
//     a wave execute in order): the values are there when the next step starts;
//   * a lane's local time is congruent to the unrolled copy's index mod 3 (there: mod 6), so a mirror rule has up to two
//     candidate values per copy;
//   * the wave's role (first / last of its workgroup: device-scope loads / stores) is read at run time: ONE copy of the
//     15-step body (three would be more than the instruction cache two CUs share; measured with three: no faster).
// Memory traffic per step is lws_skew.hip's (one row of S in, one out, one row of |S| in); steps per utterance and sweep:
// 1277 against 1792.  Same sums over the same values in the same order as lws_skew_kernel.
#include <math.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "avsi_common.h"
#include "lws_shared.h"
#include "lws_skew_weights.h"

namespace {

constexpr int KB = AVSI_LWS_KB, LMAX = AVSI_LWS_LMAX, NP = AVSI_LWS_NP, MAX_SWEEPS = AVSI_LWS_MAX_SWEEPS;
constexpr int LANES = 64, LPU = 32, UPW = LANES / LPU;      // lanes per utterance, utterances per wave
constexpr int SKEW = 9;                    // bins a frame runs behind the frame before it
constexpr int PERIOD = SKEW * LPU;         // 288 steps between two frames of a lane
constexpr int UNR = 15;                    // steps per unrolled body = ring length
constexpr int PART = 3, PARTS = UNR / PART;                // progress is published in parts of a body
constexpr int T0 = -30;                    // time of step 0 (a multiple of 15, hence of 3)
constexpr int ROW_OFF = 5;                 // row of (frame m, position x) = x + 9 m + ROW_OFF
constexpr int LOOK = UNR;                  // the old-value ring holds positions tau + 1 .. tau + LOOK
constexpr int UP_AHEAD = SKEW + 1;         // `up` of a step is the one of the neighbour's bin tau + 10 (its NEXT step)
constexpr int DN_LAG = SKEW - 1;           // `dn` of a step is the one of position tau - 8 (the neighbour's NEXT step)
constexpr int TAU_LAST = 2 * (KB - 1) - (KB - 1 - LMAX);   // 261: last mirror position above Nyquist (kept in the layout)
constexpr int TAU_END = KB - 1 + DN_LAG;   // 264: the step that makes `dn` of bin 256
constexpr int PF = 5;                      // steps between the request of a row and its use (the landing ring has UNR slots)
constexpr int OPS = 3;                     // memory operations per step and lane
// a stage in part h loads rows up to t + PF + LOOK + ROW_OFF, which its predecessor stores PF + LOOK steps later
constexpr int AHEAD = (PART - 1 + PF + LOOK) / PART + 1;
// Inside a workgroup a stage hands its rows to the next one through a ring of RING_S rows in LDS (slot = row mod 15 = a
// constant of the unrolled copy), requested PFL steps ahead.  Counters there count parts ISSUED (LDS operations of a CU are
// performed in the order they are issued: a row is there when the counter written behind it is): a stage starts part h
// when its predecessor has issued h + AHEAD_L parts, and part g when its successor has issued g - BACK (the slots it is
// about to overwrite have been read) -- a lead of 7 .. 9 parts.
constexpr int RING_S = UNR, PFL = 2;
constexpr int AHEAD_L = (PART - 1 + PFL + LOOK) / PART + 1;
constexpr int BACK = (RING_S + PFL + LOOK) / PART - 1;             // = ceil((RING_S + PFL + LOOK - (PART - 1)) / PART) - 1
constexpr int QSLACK = 6;
constexpr int CTR_STRIDE = 64;
constexpr int GPROG_INTS = (MAX_SWEEPS / 4) * CTR_STRIDE;
static_assert(TAU_LAST == 261 && KB == 257 && LMAX == 5, "the edge rules below are written out for 257 bins, L = 5");
static_assert(SKEW % 3 == 0 && UNR % 3 == 0 && T0 % UNR == 0, "a lane's local time must be congruent to the unrolled copy's index mod 3");
static_assert(UP_AHEAD + LMAX == LOOK && DN_LAG + LMAX < UNR, "the windows of the two neighbour sums must lie inside the rings");
static_assert(PERIOD >= TAU_END + 1 + (LOOK - 1), "a lane must be done with a frame before position 1 of its next one arrives");
static_assert(AHEAD == 8 && OPS * PART < 64, "counted waits");
static_assert(AHEAD_L == 7 && BACK == 9 && BACK >= AHEAD_L + 1, "the lead a stage may have over the next one inside a workgroup");

__host__ __device__ constexpr int duo_rows(int M) {         // + look-ahead of the last steps, + the scratch row
    return ((SKEW * (M - 1) + TAU_END + ROW_OFF + 1 + 2 * UNR + LOOK + PF + 8 + UNR - 1) / UNR) * UNR;
}
__host__ __device__ constexpr int duo_steps(int M) { return SKEW * (M - 1) + TAU_END - T0 + 1; }

__device__ __forceinline__ float2 cmadd(float2 acc, float wr, float wi, float2 x) {
    // (a weight that is real at compile time: the two products with +-0 add nothing but the sign of a zero)
    if (__builtin_constant_p(wi) && wi == 0.f) return make_float2(fmaf(wr, x.x, acc.x), fmaf(wr, x.y, acc.y));
    return make_float2(fmaf(wr, x.x, fmaf(-wi, x.y, acc.x)), fmaf(wr, x.y, fmaf(wi, x.x, acc.y)));
}
__device__ __forceinline__ float2 conjf2(float2 v) { return make_float2(v.x, -v.y); }
__device__ __forceinline__ float2 sel(bool c, float2 a, float2 b) { return make_float2(c ? a.x : b.x, c ? a.y : b.y); }

// (inline-asm memory operations with SGPR base pointers and counted waits: see lws_skew.hip for why, and for the `s_nop 4`)
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void row_load8(bool DEV, float2& dst, const float2* row, unsigned lane_off) {
    if (DEV) asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 sc0 sc1" : "=v"(dst) : "v"(lane_off), "s"(row) : "memory");
    else asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(row) : "memory");
}
__device__ __forceinline__ void row_load4(float& dst, const float* row, unsigned lane_off) {
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(row) : "memory");
}
__device__ __forceinline__ void row_store8(bool DEV, float2* row, unsigned lane_off, float2 v) {
    if (DEV) asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(lane_off), "v"(v), "s"(row) : "memory");
    else asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(lane_off), "v"(v), "s"(row) : "memory");
}

// every value c in [LO, HI] a lane's local time tau can take in copy I of the unrolled body: tau = t - 9 m and
// t = T0 + 12 body + I, so tau = I (mod 3) -- an edge rule is "compare tau with these constants, then move between two
// registers known at compile time"; f(integral_constant<c>) is called for each of them
template <int I, int C, class F>
__device__ __forceinline__ void cand_call(F& f) {
    if constexpr (((C - I) % 3 + 3) % 3 == 0) f(std::integral_constant<int, C>{});
}
template <int I, int LO, class F, int... K>
__device__ __forceinline__ void for_cands_impl(F& f, std::integer_sequence<int, K...>) {
    (cand_call<I, LO + K>(f), ...);
}
template <int I, int LO, int HI, class F>
__device__ __forceinline__ void for_cands(F f) {
    for_cands_impl<I, LO>(f, std::make_integer_sequence<int, HI - LO + 1>{});
}
constexpr int ring(int i) { return ((i % UNR) + UNR) % UNR; }
constexpr unsigned ROW8 = LANES * sizeof(float2), ROW4 = LANES * sizeof(float);

struct LaneState {
    float2 R[UNR], P[UNR], Ls[UNR], Lq[UNR];   // Ls / Lq: rows on their way from memory / from the predecessor's LDS ring
    float La[UNR];
    float2 cnext, unext, dnext;     // c(tau), up and dn of the coming step, requested from LDS a step ahead
    int tau, m;
};

struct StepCtx {
    const float2* Sg;               // this pair's [rows][64] complex array
    const float* Ag;                // ... and its magnitudes
    const float2* ctab;             // LDS: c(k), k mod 64
    float2* xch;                    // LDS: this wave's exchange rows, [2][64]: up, dn
    float2* ring_out;               // LDS: this stage's row ring [RING_S][64] (read by the next stage of the workgroup)
    const float2* ring_in;          // LDS: the ring of the stage before (its own for the first wave: not used there)
    int trash_row;
    bool past_only, din, dout;      // (wave-uniform)
    float thr;                      // per lane: the threshold of the lane's utterance
    int Ml;                         // per lane: frames of the lane's utterance (0: no such utterance)
    unsigned lane8, lane4;
    int lane, prev, next;           // this lane; the lanes of the frame before / after (inside the utterance's half)
};

__device__ __forceinline__ unsigned rowpos(int r) { return (unsigned)(r > 0 ? r : 0); }

// One step of a lane: bin tau of its frame.  C.din / C.dout: device-scope loads (the stage before sits in another workgroup)
// / stores (the stage after): two scalar branches per memory operation -- this kernel is for large batches, four waves per
// SIMD, where another wave's vector instruction covers a scalar one.
template <int I>
__device__ __forceinline__ void duo_step(LaneState& L, const StepCtx& C, int n) {
#define WBU(p, c) AVSI_LWS_STD_BU[p][c]
#define WBD(p, c) AVSI_LWS_STD_BD[p][c]
#define WB0(p, c) AVSI_LWS_STD_B0[p][c]
    const int t = T0 + n;                                  // wave-uniform
    // ---- the operands requested PF steps ago (the wait takes the landing registers as operands: lws_skew.hip)
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(L.Ls[I].x), "+v"(L.Ls[I].y), "+v"(L.La[I]) : "n"(OPS * PF - 2) : "memory");
    // (the first wave of a workgroup takes its rows from memory -- the stage before sits in another workgroup -- the others from
    // the ring of the wave before; the memory request stays in every wave's stream, at a fixed row where it is not needed,
    // so that the counted waits are the same for all)
    const float2 arr = sel(C.din, L.Ls[I], L.Lq[I]), upv0 = L.unext, dnv = L.dnext;
    const float amp = L.La[I];
    row_load8(C.din, L.Ls[ring(I + PF)], C.Sg, C.lane8 + (C.din ? rowpos(t + PF + LOOK + ROW_OFF) : 0u) * ROW8);     // position tau + 15 of step n + PF
    L.Lq[ring(I + PFL)] = C.ring_in[ring(I + PFL + LOOK + ROW_OFF) * LANES + C.lane];                  // ... of step n + PFL: row n + PFL + 20
    row_load4(L.La[ring(I + PF)], C.Ag, C.lane4 + rowpos(t + PF + ROW_OFF) * ROW4);                   // magnitude of the bin of step n + PF
    const int tau = L.tau, m = L.m;
    const float2 old = L.R[I];                                                         // position tau: its slot takes the arrival
    L.R[I] = arr;                                                                      // position tau + 15
    // positions 1 .. 5 of a row also define its mirror images -1 .. -5, which "arrived" 2 x steps earlier
    for_cands<I, 1, LMAX>([&](auto k) {
        constexpr int x = decltype(k)::value;
        L.R[ring(I - 2 * x)] = sel(tau + LOOK == x, conjf2(arr), L.R[ring(I - 2 * x)]);
    });
    // ---- three sums over this lane's own rings (slot I - e of R: position tau + 15 - e; slot I - e of P: position tau - e).
    //      `up` and `dn` are for the NEXT step of the neighbours, and nothing of this step waits for them: their second halves
    //      fill the issue slots of the dependent chain own -> t -> |t|^2 -> rsq -> scale -> select (a dozen levels, each
    //      waiting for the one before: left to itself the compiler puts all the sums first and the chain bare behind them).
    //      FENCE = nothing moves across; one multiply-add of a sum per level.
    // (the first tap of a sum: two products and one multiply-add -- a literal weight where `fma(w, x, 0)` would need an SGPR)
#define FIRST(w0, w1, val) make_float2(fmaf(w0, (val).x, -(w1) * (val).y), fmaf(w0, (val).y, (w1) * (val).x))
#define UP_TAP(p) up = cmadd(up, WBU((p) + LMAX, 0), WBU((p) + LMAX, 1), L.R[ring(I - (LMAX - (p)))])               /* tau + 10 + p */
#define DN_TAP(p) dn = cmadd(dn, WBD((p) + LMAX, 0), WBD((p) + LMAX, 1), L.P[ring(I - (DN_LAG - (p)))])            /* tau - 8 + p */
#define FENCE __builtin_amdgcn_sched_barrier(0)
    float2 up = FIRST(WBU(0, 0), WBU(0, 1), L.R[ring(I - 2 * LMAX)]);                                                             // tau + 5
    UP_TAP(-4), UP_TAP(-3), UP_TAP(-2), UP_TAP(-1);
    float2 own = FIRST(WB0(LMAX + 1, 0), WB0(LMAX + 1, 1), L.R[ring(I - (LOOK - 1))]);                                            // tau + 1
#pragma unroll
    for (int p = 2; p <= LMAX; ++p) own = cmadd(own, WB0(LMAX + p, 0), WB0(LMAX + p, 1), L.R[ring(I - (LOOK - p))]);               // tau + p
    float2 dn = FIRST(WBD(0, 0), WBD(0, 1), L.P[ring(I - (DN_LAG + LMAX))]);                                                      // tau - 13
    DN_TAP(-4), DN_TAP(-3), DN_TAP(-2), DN_TAP(-1);
#pragma unroll
    for (int p = LMAX; p >= 1; --p) own = cmadd(own, WB0(LMAX - p, 0), WB0(LMAX - p, 1), L.P[ring(I - p)]);                        // tau - p
    const float2 c = L.cnext;
    float2 upv = upv0;
    if (C.past_only) own = make_float2(0.f, 0.f), upv = own;
    const bool live = m < C.Ml;
    const bool valid = tau >= 0 && tau <= KB - 1 && live;
    FENCE;
    float2 T = own;
    T.x = fmaf(-c.y, upv.y, T.x), T.y = fmaf(c.y, upv.x, T.y);
    UP_TAP(0);
    FENCE;
    T.x = fmaf(c.x, upv.x, T.x), T.y = fmaf(c.x, upv.y, T.y);                             // c . up
    DN_TAP(0);
    FENCE;
    T.x = fmaf(c.y, dnv.y, T.x), T.y = fmaf(-c.y, dnv.x, T.y);
    UP_TAP(1);
    FENCE;
    T.x = fmaf(c.x, dnv.x, T.x), T.y = fmaf(c.x, dnv.y, T.y);                             // conj(c) . down
    DN_TAP(1);
    FENCE;
    const float yy = T.y * T.y;
    UP_TAP(2);
    FENCE;
    const float n2 = fmaf(T.x, T.x, yy);                 // (written as the fused form: the compiler's own choice differs from copy to copy)
    DN_TAP(2);
    FENCE;
    const float rs = __builtin_amdgcn_rsqf(n2);
    const bool upd = valid && amp > C.thr && n2 > 0.f;
    UP_TAP(3), DN_TAP(3);
    FENCE;
    const float sc = amp * rs;
    UP_TAP(4);
    FENCE;
    const float2 vs = make_float2(T.x * sc, T.y * sc);
    DN_TAP(4);
    FENCE;
    const float2 v = sel(upd, vs, old);
    UP_TAP(5), DN_TAP(5);
    FENCE;
    // ---- exchange for the NEXT step: this frame's `up` goes to the frame before (at its bin tau + 10), its `dn` to the frame
    //      after (at its bin tau - 8).  No select "zero where there is nothing to give": a neighbour that is at a bin
    //      (the only one that uses what it gets) finds either the right frame here, at the right position, or a lane whose rings
    //      hold nothing but zeros -- a frame >= M (its cells are never written, by any stage: what arrives is zero, what is
    //      handed on is what arrived), or the last lane of the half before its first frame has begun (frame -1).
    C.xch[C.lane] = up;
    C.xch[LANES + C.lane] = dn;
    L.unext = C.xch[C.next];
    L.dnext = C.xch[LANES + C.prev];
    L.cnext = C.ctab[(tau + 1) & 63];
#undef FIRST
#undef UP_TAP
#undef DN_TAP
#undef FENCE
    // ---- what this lane hands on for position tau: the bin, or a mirror image
    float2 out = v;
    // below DC (tau in [-5, -1]): conj of the old bins 5 .. 1, the start values of the "new" ring
    for_cands<I, -LMAX, -1>([&](auto k) {
        constexpr int tc = decltype(k)::value;
        out = sel(tau == tc, conjf2(L.R[ring(I - (LOOK + 2 * tc))]), out);
    });
    // above Nyquist (tau in [257, 261]): conj of the NEW bins 255 .. 251
    for_cands<I, KB, TAU_LAST>([&](auto k) {
        constexpr int tc = decltype(k)::value;
        out = sel(tau == tc, conjf2(L.P[ring(I - (2 * tc - 2 * (KB - 1)))]), out);
    });
    L.P[I] = out;
    // bins 1 .. 5 refresh their mirror images below DC (2 x steps back in the "new" ring) ...
    for_cands<I, 1, LMAX>([&](auto k) {
        constexpr int x = decltype(k)::value;
        L.P[ring(I - 2 * x)] = sel(tau == x, conjf2(v), L.P[ring(I - 2 * x)]);
    });
    // ... and bins 251 .. 255 theirs above Nyquist (positions 512 - tau of the "old" ring, still ahead of this lane)
    for_cands<I, KB - 1 - LMAX, KB - 2>([&](auto k) {
        constexpr int tc = decltype(k)::value;
        constexpr int e = LOOK - (2 * (KB - 1) - 2 * tc);            // slots back from the newest arrival
        L.R[ring(I - e)] = sel(tau == tc, conjf2(v), L.R[ring(I - e)]);
    });
    // ---- store (row t + 5, every lane, always: lanes with nothing to store write a scratch row)
    const bool st = tau >= 0 && tau <= TAU_LAST && live;
    // (to the next stage: through this stage's ring, row n + 5; to memory only where the next stage sits in another workgroup)
    C.ring_out[ring(I + ROW_OFF) * LANES + C.lane] = out;
    const unsigned off = C.lane8 + (unsigned)(st && C.dout ? t + ROW_OFF : C.trash_row) * ROW8;
    row_store8(C.dout, const_cast<float2*>(C.Sg), off, out);
    // ---- next step of this lane
    const int nt = tau + 1;
    const bool wrap = nt > TAU_END;
    L.tau = wrap ? nt - PERIOD : nt;
    L.m = wrap ? m + LPU : m;
    __builtin_amdgcn_sched_barrier(0);          // steps are not interleaved: their live ranges would add up
#undef WBU
#undef WBD
#undef WB0
}

template <int H, int... Is>
__device__ __forceinline__ void duo_part(LaneState& L, const StepCtx& C, int n0, std::integer_sequence<int, Is...>) {
    (duo_step<H * PART + Is>(L, C, n0 + H * PART + Is), ...);
}

struct StageCtx {
    int wv, wg, lane, stage, stages, nb, hlast, nbp;
    int* gprog;
    volatile int* vprog;
    int* status;
    float mean;                 // per lane: the mean magnitude of the lane's utterance
    float mean2[UPW], amax2[UPW];
};

// What a stage carries from one pair of utterances to the next one of its workgroup (a workgroup runs the pairs slot, slot +
// slots, ... one after the other, every wave going on to the next pair as soon as it has finished its last sweep of this one: the
// partial last round of a pair and the filling of the pipeline overlap with the neighbour pair's sweeps).  The counters of the
// waves are monotone over ALL sweeps a stage has run, so a stage needs how many sweeps its two neighbour stages have behind them.
struct Carry {
    int known, known_s, mine, prev_succ_done, done_pred, done_succ;
    bool dead;
};

// the sweeps of one stage over one pair (lws_skew.hip: skew_sweeps; here with the two-way hand-shake of the LDS rings)
template <int NW>
__device__ __forceinline__ void duo_sweeps(StepCtx& C, const AvsiLwsSchedule& sched, const StageCtx& Q, Carry& K) {
    const int wv = Q.wv, wg = Q.wg, lane = Q.lane, stage = Q.stage, stages = Q.stages, nb = Q.nb, hlast = Q.hlast, nbp = Q.nbp;
    const int total = PARTS * nb;                 // parts of a sweep
    int* gprog = Q.gprog;
    volatile int* vprog = Q.vprog;                // parts ISSUED, per wave of this workgroup (monotone over the sweeps of a stage)
    int* status = Q.status;
    const bool dev_in = C.din, dev_out = C.dout;
    bool dead = K.dead;
    int known = dev_in ? 0 : K.known, known_s = K.known_s, mine = K.mine;      // (a first wave polls this pair's own counters in memory)
    int n_active = 0;
    for (int sw = 0; sw < sched.n; ++sw) {
        bool active = false;
#pragma unroll
        for (int u = 0; u < UPW; ++u) active = active || Q.amax2[u] > sched.rel[sw] * Q.mean2[u];
        n_active += active ? 1 : 0;
    }
    int rank_a = -1, last_active = -1;
    int prev_succ_done = K.prev_succ_done;        // counter value of the next wave that says "has read all of my last sweep"
    for (int sw = 0; sw < sched.n; ++sw) {
        const float rel = sched.rel[sw];
        bool active = false;
#pragma unroll
        for (int u = 0; u < UPW; ++u) active = active || Q.amax2[u] > rel * Q.mean2[u];
        if (!active) continue;                    // idle for both: changes nothing, needs no stage
        const int pred = last_active;
        last_active = sw, ++rank_a;
        if (rank_a % stages != stage) continue;
        C.thr = rel * Q.mean;
        C.past_only = sched.past_only[sw] != 0;
        const int pstage = pred < 0 ? -1 : (rank_a - 1) % stages;
        const int pbase = pred < 0 ? 0 : (K.done_pred + (rank_a - 1) / stages) * nbp;
        const int base = mine * nbp;
        // the sweep after this one runs on the next wave of this workgroup (unless this is the last wave, or the last sweep)
        const bool succ_here = !dev_out && rank_a + 1 < n_active;
        C.dout = dev_out || rank_a + 1 >= n_active;           // rows go to memory for another workgroup -- or as the result
        const int sbase = (K.done_succ + (rank_a + 1) / stages) * nbp;
        LaneState L;
#pragma unroll
        for (int i = 0; i < UNR; ++i) L.R[i] = L.P[i] = L.Ls[i] = L.Lq[i] = make_float2(0.f, 0.f), L.La[i] = 0.f;
        L.unext = L.dnext = make_float2(0.f, 0.f);
        L.m = lane & (LPU - 1);
        L.tau = T0 - SKEW * (lane & (LPU - 1));
        L.cnext = C.ctab[L.tau & 63];
        auto spin = [&](auto load, int want, int& seen) {        // bounded: a stage that never comes sets the status word
            int spins = 0;
            for (;;) {
                seen = __builtin_amdgcn_readfirstlane(load());
                if (seen >= want) break;
                if (seen + 4 < want) __builtin_amdgcn_s_sleep(64);
                else __builtin_amdgcn_s_sleep(4);
                if (++spins > (1 << 22)) {
                    dead = true;
                    if (lane == 0 && status) atomicOr(status, 1);
                    break;
                }
            }
        };
        auto wait_pred = [&](int part) {           // part: index of the part about to start
            if (pstage < 0 || dead) return;
            if (dev_in) {                          // another workgroup: parts COMPLETE, in device memory
                const int need = pbase + (part + AHEAD < hlast ? part + AHEAD : hlast);
                if (known >= need) return;
                const int want = need + QSLACK < pbase + hlast ? need + QSLACK : pbase + hlast;
                spin([&] { return __hip_atomic_load(gprog + (pstage / NW) * CTR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, want, known);
            } else {                               // the wave before: parts ISSUED, in LDS
                const int need = pbase + (part + AHEAD_L < total ? part + AHEAD_L : total);
                if (known >= need) return;
                spin([&] { return vprog[wv - 1]; }, need, known);
            }
        };
        auto wait_succ = [&](int part) {           // the slots this part overwrites have been read by the wave after
            if (!succ_here || dead || part < BACK) return;
            const int need = sbase + part - BACK;
            if (known_s >= need) return;
            spin([&] { return vprog[wv + 1]; }, need, known_s);
        };
        auto issued = [&](int parts) {             // (behind the rows of these parts: LDS operations keep their order)
            asm volatile("" ::: "memory");
            if (lane == 0) vprog[wv] = base + parts;
        };
        auto completed = [&](int value) {          // (the last wave of a workgroup: its rows went to memory)
            if (lane == 0) __hip_atomic_store(gprog + wg * CTR_STRIDE, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // the ring is about to be written from its start again: the wave after must be through with the sweep before
        if (prev_succ_done >= 0 && !dead) spin([&] { return vprog[wv + 1]; }, prev_succ_done, known_s);
        // prefetch: the rows of steps 0 .. PF - 1, with the step loop's three operations per (virtual) step
        wait_pred(0);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            row_load8(dev_in, L.Ls[i], C.Sg, C.lane8 + (dev_in ? rowpos(T0 + i + LOOK + ROW_OFF) : 0u) * ROW8);
            row_load4(L.La[i], C.Ag, C.lane4 + rowpos(T0 + i + ROW_OFF) * ROW4);
            row_store8(false, const_cast<float2*>(C.Sg), C.lane8 + (unsigned)C.trash_row * ROW8, make_float2(0.f, 0.f));
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < PFL; ++i) L.Lq[i] = C.ring_in[ring(i + LOOK + ROW_OFF) * LANES + lane];
        constexpr std::make_integer_sequence<int, PART> seq{};
        auto part = [&](auto k, int body) {
            constexpr int K = decltype(k)::value;
            const int idx = PARTS * body + K;
            wait_pred(idx);
            wait_succ(idx);
            asm volatile("" ::: "memory");
            duo_part<K>(L, C, body * UNR, seq);
            issued(idx + 1);
            if (dev_out) {
                // everything issued before this part's 9 operations is complete: the stores of the parts before it
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS * PART) : "memory");
                completed(base + idx);
            }
        };
        for (int body = 0; body < nb; ++body) {
            part(std::integral_constant<int, 0>{}, body);
            part(std::integral_constant<int, 1>{}, body);
            part(std::integral_constant<int, 2>{}, body);
            part(std::integral_constant<int, 3>{}, body);
            part(std::integral_constant<int, 4>{}, body);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the requests of the last steps land in registers nobody reads any more: reserved until they have landed)
#pragma unroll
        for (int i = 0; i < PF; ++i) asm volatile("" : "+v"(L.Ls[i].x), "+v"(L.Ls[i].y), "+v"(L.La[i]));
        if (dev_out) completed(base + hlast);
        prev_succ_done = succ_here ? sbase + total : -1;
        ++mine;
    }
    // sweeps of this pair that ran on the stage before / after this one
    auto ran_on = [&](int st) { return n_active > st ? (n_active - st - 1) / stages + 1 : 0; };
    K.done_pred += ran_on((stage + stages - 1) % stages), K.done_succ += ran_on((stage + 1) % stages);
    K.known = known, K.known_s = known_s, K.mine = mine, K.prev_succ_done = prev_succ_done, K.dead = dead;
}

// NW waves per workgroup = NW pipeline stages of a pair of utterances; G workgroups per pair; `slots` pairs at a time, the
// workgroups of slot k running the pairs k, k + slots, ... one after the other
template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void lws_duo_kernel(float2* __restrict__ Sall, const float* __restrict__ Aall, int B, int M,
                                                         const AvsiLwsSchedule sched, int* __restrict__ status,
                                                         const float2* __restrict__ stats, int* __restrict__ gprog_all, int G,
                                                         int phase_step, int pairs, int slots) {
    __shared__ float2 ctab[64];
    __shared__ int prog[NW];
    __shared__ float2 xch[NW][2 * LANES];
    __shared__ float2 srings[NW][RING_S * LANES];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int slot = blockIdx.x / G, wg = blockIdx.x - slot * G;
    const int stage = wg * NW + wv, stages = G * NW;
    const int rows = duo_rows(M);
    const int nb = (duo_steps(M) + UNR - 1) / UNR;
    const int hlast = PARTS * nb + AHEAD + 1;                 // published when a sweep is finished: every need is capped there
    const int nbp = PARTS * nb + 16;                          // counter values per sweep of a stage
    volatile int* vprog = prog;
    if (threadIdx.x < 64) {
        float sn, cs;
        sincospif(-2.f * (float)((threadIdx.x * phase_step) & 63) / 64.f, &sn, &cs);
        ctab[threadIdx.x] = make_float2(cs, sn);
    }
    if (threadIdx.x < NW) prog[threadIdx.x] = 0;
    xch[wv][lane] = xch[wv][LANES + lane] = make_float2(0.f, 0.f);
    __syncthreads();
    StageCtx Q;
    StepCtx C;
    C.ctab = ctab;
    C.xch = xch[wv];
    C.ring_out = srings[wv];
    C.ring_in = srings[wv > 0 ? wv - 1 : 0];
    C.trash_row = rows - 1;
    C.lane8 = lane * 8u, C.lane4 = lane * 4u;
    C.lane = lane;
    C.prev = (lane & ~(LPU - 1)) | ((lane - 1) & (LPU - 1));
    C.next = (lane & ~(LPU - 1)) | ((lane + 1) & (LPU - 1));
    Q.wv = wv, Q.wg = wg, Q.lane = lane, Q.stage = stage, Q.stages = stages, Q.nb = nb, Q.hlast = hlast, Q.nbp = nbp;
    Q.vprog = vprog, Q.status = status;
    Carry K{0, 0, 0, -1, 0, 0, false};
    for (int q = slot; q < pairs; q += slots) {
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int b = q * UPW + u;
            const float2 st = b < B ? stats[b] : make_float2(0.f, 0.f);
            Q.mean2[u] = st.x, Q.amax2[u] = st.y;
        }
        const int bl = q * UPW + lane / LPU;
        Q.mean = bl < B ? stats[bl].x : 0.f;
        Q.gprog = gprog_all + (size_t)q * GPROG_INTS;
        const size_t cells = (size_t)q * rows * LANES;
        C.Sg = uniform_ptr(Sall + cells);
        C.Ag = uniform_ptr(Aall + cells);
        C.Ml = bl < B ? M : 0;
        // the stage before the first wave / after the last one sits in another workgroup (or is this one in the next round)
        C.din = wv == 0, C.dout = wv == NW - 1;
        duo_sweeps<NW>(C, sched, Q, K);
    }
}

// spec [B][M][257] -> the diagonal layout of a pair (bins, the five mirror positions above Nyquist, magnitudes); everything
// else zero, every cell written.  A block owns sixteen lanes (frames 32 rnd + 16 h + 0 .. 15 of one utterance) and, of each
// lane, the 288 rows of its round: the frames come in through LDS in whole spectra (coalesced along the bins) and go out in
// whole 128-byte pieces of rows (coalesced along the lanes) -- a thread per cell read or wrote 8 bytes at a 2 KB stride
// (0.5 ms each way at 1024 utterances).  The first round's blocks also write the rows above a lane's first frame, the last
// round's the rows below its last one (and the scratch row).
constexpr int TL = 16;                          // lanes per block of the two layout kernels
__device__ __forceinline__ void layout_span(int rnd, int h, int rounds, int rows, int& rlo, int& rhi) {
    rlo = rnd == 0 ? 0 : PERIOD * rnd + ROW_OFF + SKEW * TL * h;
    rhi = rnd == rounds - 1 ? rows : PERIOD * (rnd + 1) + ROW_OFF + SKEW * (TL * h + TL - 1) + 1;
}

__global__ __launch_bounds__(256) void lws_to_duo_kernel(const float2* __restrict__ spec, int B, int M, int rows, int rounds,
                                                        float2* __restrict__ S, float* __restrict__ A) {
    __shared__ float2 tile[TL][KB];
    const int tid = threadIdx.x, rnd = blockIdx.x >> 1, h = blockIdx.x & 1, b = blockIdx.y, q = b / UPW, u = b - q * UPW;
    const int m0 = rnd * LPU + TL * h;
    for (int i = tid; i < TL * KB; i += 256) {
        const int fr = i / KB, x = i - fr * KB, m = m0 + fr;
        tile[fr][x] = (b < B && m < M) ? spec[((int64_t)b * M + m) * KB + x] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    int rlo, rhi;
    layout_span(rnd, h, rounds, rows, rlo, rhi);
    const int jj = tid & (TL - 1), j = TL * h + jj, m = m0 + jj;
    const bool live = b < B && m < M;
    for (int r = rlo + (tid >> 4); r < rhi; r += 256 / TL) {
        const int x = r - ROW_OFF - SKEW * j - PERIOD * rnd;             // position inside this round's frame of the lane
        if ((x < 0 && rnd > 0) || (x >= PERIOD && rnd < rounds - 1)) continue;          // another round's rows of this lane
        float2 v = make_float2(0.f, 0.f);
        float a = 0.f;
        if (live && x >= 0 && x <= TAU_LAST && r < rows - 1) {
            if (x < KB) {
                v = tile[jj][x];
                a = sqrtf(fmaf(v.x, v.x, v.y * v.y));      // (the fused form, spelled out: as lws_skew.hip)
            } else {
                v = conjf2(tile[jj][2 * (KB - 1) - x]);
            }
        }
        const int64_t at = ((int64_t)q * rows + r) * LANES + u * LPU + j;
        S[at] = v;
        A[at] = a;
    }
}

__global__ __launch_bounds__(256) void lws_from_duo_kernel(const float2* __restrict__ S, int B, int M, int rows, int rounds,
                                                          float2* __restrict__ spec) {
    __shared__ float2 tile[TL][KB];
    const int tid = threadIdx.x, rnd = blockIdx.x >> 1, h = blockIdx.x & 1, b = blockIdx.y, q = b / UPW, u = b - q * UPW;
    if (b >= B) return;
    const int m0 = rnd * LPU + TL * h;
    const int jj = tid & (TL - 1), j = TL * h + jj;
    const int rlo = PERIOD * rnd + ROW_OFF + SKEW * TL * h, rhi = PERIOD * rnd + ROW_OFF + SKEW * (TL * h + TL - 1) + KB;
    for (int r = rlo + (tid >> 4); r < rhi; r += 256 / TL) {
        const int x = r - ROW_OFF - SKEW * j - PERIOD * rnd;
        if (x >= 0 && x < KB && r < rows - 1) tile[jj][x] = S[((int64_t)q * rows + r) * LANES + u * LPU + j];
    }
    __syncthreads();
    for (int i = tid; i < TL * KB; i += 256) {
        const int fr = i / KB, x = i - fr * KB, m = m0 + fr;
        if (m < M) spec[((int64_t)b * M + m) * KB + x] = tile[fr][x];
    }
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// The launch shape for `duos` pairs of utterances: NW stages per workgroup (16 unless given), G workgroups per pair so that
// the pairs of a launch fill the chip (16 waves per CU: 128 VGPRs) without more stages than sweeps.
bool duo_shape(int duos, int sweeps, int& nw, int& g) {
    if (nw == 0) nw = 16;
    const int cap = AVSI_NUM_CU * (16 / nw);
    if (g == 0) {
        g = cap / duos;
        const int most = (sweeps + nw - 1) / nw;
        if (g > most) g = most;
        if (g < 1) g = 1;
    }
    return g >= 1 && g * nw <= MAX_SWEEPS && g <= MAX_SWEEPS / 4 && g <= cap;
}

}  // namespace

// word 0: status; (mean, max) per utterance; one row of stage counters per pair; S (complex) and |S| per pair
extern "C" size_t avsi_lws_run_duo_workspace_bytes(int batch, int num_frames) {
    if (batch <= 0 || num_frames <= 0) return 0;
    const size_t duos = ((size_t)batch + UPW - 1) / UPW;
    const size_t cells = duos * duo_rows(num_frames) * LANES;
    return align256(16 + (size_t)batch * sizeof(float2)) + align256(duos * GPROG_INTS * sizeof(int)) + align256(cells * sizeof(float2)) +
           align256(cells * sizeof(float));
}

extern "C" int avsi_lws_run_duo_f32(float* spec, int batch, int num_frames, int frame_len, int hop, int nfft, int L,
                                     int nofuture_iterations, float nofuture_alpha, int online_iterations, float online_alpha,
                                     int batch_iterations, float batch_alpha, float batch_beta, float batch_gamma,
                                     int waves_per_group, int groups_per_pair, void* workspace, size_t workspace_bytes, void* stream) {
    if (!spec || batch <= 0 || num_frames <= 0 || L < 1 || nofuture_iterations < 0 || online_iterations < 0 || batch_iterations < 0)
        return AVSI_ERR_INVALID_ARG;
    AvsiLwsSchedule S;
    if (!avsi_lws_geometry_ok(frame_len, hop, nfft) || L > LMAX ||
        !avsi_lws_make_schedule(nofuture_iterations, nofuture_alpha, online_iterations, online_alpha, batch_iterations, batch_alpha,
                                batch_beta, batch_gamma, S))
        return AVSI_ERR_UNSUPPORTED;
    // the reference's geometry only (weights as compile-time constants, checked against the ones the library computes):
    // other geometries take avsi_lws_run_skew_f32
    if (frame_len != AVSI_LWS_STD_FRAME || hop != AVSI_LWS_STD_HOP || nfft != AVSI_LWS_STD_NFFT || L != AVSI_LWS_STD_L)
        return AVSI_ERR_UNSUPPORTED;
    {
        double alpha[3][NP][2];
        avsi_lws_host_alpha(frame_len, hop, nfft, L, alpha);
        for (int p = -LMAX; p <= LMAX; ++p) {
            const double ph = -2.0 * M_PI * p * hop / nfft, cs = cos(ph), sn = sin(ph);
            const double *au = alpha[2][p + LMAX], *ad = alpha[0][p + LMAX], *a0 = alpha[1][p + LMAX];
            const float w[6] = {(float)(au[0] * cs - au[1] * sn), (float)(au[0] * sn + au[1] * cs), (float)(ad[0] * cs + ad[1] * sn),
                                (float)(-ad[0] * sn + ad[1] * cs), (float)a0[0], (float)a0[1]};
            const float* k[3] = {AVSI_LWS_STD_BU[p + LMAX], AVSI_LWS_STD_BD[p + LMAX], AVSI_LWS_STD_B0[p + LMAX]};
            for (int i = 0; i < 6; ++i)
                if (fabs((double)w[i] - k[i / 2][i % 2]) > 1e-7) return AVSI_ERR_UNSUPPORTED;      // lws_skew_weights.h is stale
        }
    }
    if (S.n == 0) return AVSI_OK;
    if (!workspace || workspace_bytes < avsi_lws_run_duo_workspace_bytes(batch, num_frames)) return AVSI_ERR_WORKSPACE;
    if (waves_per_group != 0 && waves_per_group != 4 && waves_per_group != 8 && waves_per_group != 16) return AVSI_ERR_INVALID_ARG;
    if (groups_per_pair < 0) return AVSI_ERR_INVALID_ARG;
    const int duos = (batch + UPW - 1) / UPW;
    const hipStream_t st = (hipStream_t)stream;
    const int rows = duo_rows(num_frames);
    const size_t cells = (size_t)duos * rows * LANES;
    char* ws = static_cast<char*>(workspace);
    int* status = reinterpret_cast<int*>(ws);
    float2* stats = reinterpret_cast<float2*>(ws + 16);
    size_t off = align256(16 + (size_t)batch * sizeof(float2));
    int* gprog = reinterpret_cast<int*>(ws + off);
    off += align256((size_t)duos * GPROG_INTS * sizeof(int));
    float2* Sd = reinterpret_cast<float2*>(ws + off);
    off += align256(cells * sizeof(float2));
    float* Ad = reinterpret_cast<float*>(ws + off);
    if (hipMemsetAsync(workspace, 0, align256(16 + (size_t)batch * sizeof(float2)) + align256((size_t)duos * GPROG_INTS * sizeof(int)), st) !=
        hipSuccess)
        return AVSI_ERR_LAUNCH;
    avsi_clear_error();
    avsi_lws_launch_stats(spec, batch, num_frames, reinterpret_cast<float*>(stats), st);
    const int phase_step = 64 * hop / nfft;
    float2* sp = reinterpret_cast<float2*>(spec);
    // every workgroup of a launch must be resident (its stages wait for each other): the chip holds `slots` pairs at a time,
    // and the workgroups of a slot run their pairs one after the other inside the kernel
    int NW = waves_per_group, G = groups_per_pair;
    if (!duo_shape(duos, S.n, NW, G)) return AVSI_ERR_INVALID_ARG;
    int slots = AVSI_NUM_CU * (16 / NW) / G;
    if (const char* e = getenv("AVSI_LWS_DUO_SLOTS"))          // (tests: several pairs per workgroup at small batches)
        if (atoi(e) > 0 && atoi(e) < slots) slots = atoi(e);
    if (slots > duos) slots = duos;
    const int rounds = (num_frames + LPU - 1) / LPU;
    hipLaunchKernelGGL(lws_to_duo_kernel, dim3(2 * rounds, duos * UPW), dim3(256), 0, st, sp, batch, num_frames, rows, rounds, Sd, Ad);
#define AVSI_DUO_LAUNCH(NWV)                                                                                                        \
    hipLaunchKernelGGL((lws_duo_kernel<NWV>), dim3(slots* G), dim3(64 * (NWV)), 0, st, Sd, Ad, batch, num_frames, S, status, stats, \
                       gprog, G, phase_step, duos, slots)
    if (NW == 16) AVSI_DUO_LAUNCH(16);
    else if (NW == 8) AVSI_DUO_LAUNCH(8);
    else AVSI_DUO_LAUNCH(4);
#undef AVSI_DUO_LAUNCH
    hipLaunchKernelGGL(lws_from_duo_kernel, dim3(2 * rounds, duos * UPW), dim3(256), 0, st, Sd, batch, num_frames, rows, rounds, sp);
    return avsi_launch_status();
}
