// LWS sweeps with the FRAMES of a sweep in the lanes of a wave (skewed-frame pipeline) -- the default kernel behind
// avsi_lws_run_skew_f32, the phase refinement of the reference's `infer` (inference.py:119,141-154; see lws.hip for the
// algorithm, the conventions and the transforms around it).
//
// A sweep updates S[m, k] <- A[m, k] t / |t| in raster order IN PLACE, t = sum over rows m-1, m, m+1 and |p| <= 5 bins.
// Bin (m, k) needs (m-1, k+5) and (m, k-1) of THIS sweep and (m, k+5), (m+1, k+5) of the sweep before, nothing else:
//   * frame m may run LMAX + 1 = 6 bins behind frame m-1            -> the 64 lanes of a wave work on 64 consecutive frames,
//     lane j at bin k = t - 6 m at step t (lane = m mod 64; a lane is busy 257 + margins of every 384 steps);
//   * sweep s may run a few dozen steps behind sweep s-1            -> the sweeps of an utterance are a pipeline of waves.
// lws_sweeps_kernel (lws.hip) ran the in-frame recurrence on ONE lane (four with four utterances per wave): 60 of 64
// lanes idle, 16 us per frame whatever else happened.  Here every lane runs a recurrence.
//
// What a lane keeps (all in registers, rings of 12 indexed by step mod 12, so every index is a compile-time constant
// in the 12-fold unrolled step loop):
//   R  its own row's values from the sweep before ("old"), positions k .. k+11, streamed from memory PF = 6 steps ahead;
//   P  its own row's values of this sweep ("new"), positions k-1 .. k-12.
// Per step it forms three sums over ITS OWN rings and hands two of them to its neighbours with one DPP rotate each:
//   own  = sum_{p=1..5} a_0(p) R[k+p] + a_0(-p) P[k-p]                  (10 taps, its own bin)
//   up   = sum_p b_{+1}(p) R[k+6+p]   for lane j-1, whose bin is k+6     (row m+1 of that lane: still the old values)
//   down = sum_p b_{-1}(p) P[k-6+p]   for lane j+1, whose bin is k-6     (row m-1 of that lane: already the new values)
//   t = own + c(k) . up(from j+1) + conj c(k) . down(from j-1),          c(k) = e^{-2 pi j k R / N},
//   b_q(p) = alpha_q(p) c_q(p): the consistency phase factor e^{-2 pi j (k+p) q R / N} splits into a constant weight and
//   one per-bin rotation.  The loop-carried chain per step is one complex MAC (the newest tap), the rotate, the norm,
//   a reciprocal square root and two multiplies; the other ~30 MACs are independent of it.
// Conjugate-mirror bins (k < 0, k > 256) are positions of the same rings: the five below DC are written when their
// source arrives / changes (predicated register moves), the five above Nyquist are stored in the layout like bins.
//
// Memory layout ("diagonal-major"): element (m, x), x in [-5, 261], lives in row x + 6 m + 5, column m mod 64 of a
// [rows][64] complex array per utterance -- at step t EVERY lane loads row t + 16 + PF (its own position k + 11, PF = 6
// steps ahead) and stores row t + 5 (its own bin): two 512-byte rows per step, nothing strided.  Magnitudes (constant
// through all sweeps, as in oracle/lws.py) sit in a float array of the same shape.
//
// Pipeline of sweeps: a workgroup is NW waves = NW consecutive (active) sweeps of ONE utterance, G workgroups chain up
// per utterance (G * NW stages; the a-th active sweep runs on stage a mod stages).  A stage publishes "half bodies (6 steps)
// finished and visible" -- in LDS for the next wave of its workgroup, which reads the rows back through the CU's own L1
// (coherent within a workgroup), in global memory (device-scope stores / loads) where the next stage sits in another
// workgroup or is wave 0 of the next round.  A stage starts a half body (6 steps) when its predecessor has published
// AHEAD half bodies more (the exact data dependence, see AHEAD).
// All stages of an utterance must be resident together (bounded waits, status word).
#include <math.h>

#include <utility>

#include "avsi_common.h"
#include "lws_shared.h"
#include "lws_skew_weights.h"

namespace {

constexpr int KB = AVSI_LWS_KB, LMAX = AVSI_LWS_LMAX, NP = AVSI_LWS_NP, MAX_SWEEPS = AVSI_LWS_MAX_SWEEPS;
constexpr int SKEW = LMAX + 1;             // bins a frame runs behind the frame before it
constexpr int LANES = 64;
constexpr int PERIOD = SKEW * LANES;       // 384 steps between two frames of a lane
constexpr int UNR = 12;                    // steps per unrolled body = ring length = prefetch distance
constexpr int T0 = -24;                    // time of step 0 (a multiple of 12 and of 6)
constexpr int ROW_OFF = 5;                 // row of (frame m, position x) = x + 6 m + ROW_OFF
constexpr int TAU_LAST = 2 * (KB - 1) - (KB - 1 - LMAX) + 0;   // 261: last mirror position above Nyquist
constexpr int NONE = -100000;
// Progress is counted in HALF bodies (6 steps): a published value h says "the stores of the half bodies before h are
// complete".  Half x of body b (steps 12 b + 6 x ..) loads rows up to t + PF + 16, which the predecessor stores up to its
// step 12 b + 6 x + 5 + PF + 11, i.e. in its half body 2 b + x + (PF + 16) / 6: the stage starts that half once it has seen
// one more.  PF = 12 (round 3's first version, bodies as the unit and one body of slack: 60 steps between consecutive
// sweeps) -> half bodies, no slack: 36 -> PF = 6: 30 steps; 1 utterance 4.6 -> 4.0 -> 3.7 -> 3.4 ms.
constexpr int HALF = UNR / 2;
constexpr int PF = 6;                 // steps between the request of a row and its use (the landing ring has UNR slots)
constexpr int AHEAD = (PF + 16) / HALF + 1;
constexpr int QSLACK = 6;             // half bodies a stage asks for beyond its need when it has to poll device memory
constexpr int CTR_STRIDE = 64;                                        // ints between two device-scope counters (256 bytes)
constexpr int GPROG_INTS = (MAX_SWEEPS / 4) * CTR_STRIDE;            // per utterance: up to MAX_SWEEPS / 4 workgroups
static_assert(TAU_LAST == 261 && KB == 257 && SKEW == 6, "the edge rules below are written out for 257 bins, L = 5");
static_assert(LANES * SKEW >= TAU_LAST + 1 + 24, "a lane must be done with a frame (and its prefetch) before the next one starts");

struct SkewConst {          // kernel argument: the compiler keeps what it needs in SGPRs
    float bu[NP][2];        // b_{+1}(p), p = -5 .. 5
    float bd[NP][2];        // b_{-1}(p)
    float b0[NP][2];        // alpha_0(p) (the centre tap is not used)
    int phase_step;         // 64 R / N: c(k) = e^{-2 pi j k phase_step / 64}
};

// rows of the diagonal layout for M frames (a multiple of UNR; the tail covers the look-ahead of the last steps)
__host__ __device__ constexpr int skew_rows(int M) { return ((SKEW * (M - 1) + TAU_LAST + ROW_OFF + 1 + 2 * UNR + 16 + UNR - 1) / UNR) * UNR; }
__host__ __device__ constexpr int skew_steps(int M) { return SKEW * (M - 1) + TAU_LAST - T0 + 1; }

__device__ __forceinline__ float2 cmadd(float2 acc, float wr, float wi, float2 x) {
    // (a weight that is real at compile time: the two products with +-0 add nothing but the sign of a zero)
    if (__builtin_constant_p(wi) && wi == 0.f) return make_float2(fmaf(wr, x.x, acc.x), fmaf(wr, x.y, acc.y));
    return make_float2(fmaf(wr, x.x, fmaf(-wi, x.y, acc.x)), fmaf(wr, x.y, fmaf(wi, x.x, acc.y)));
}
__device__ __forceinline__ float2 conjf2(float2 v) { return make_float2(v.x, -v.y); }
__device__ __forceinline__ float2 sel(bool c, float2 a, float2 b) { return make_float2(c ? a.x : b.x, c ? a.y : b.y); }
__device__ __forceinline__ float dpp_from_prev(float v) {      // lane j gets lane j - 1 (lane 0 gets lane 63)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false));   // wave_ror:1
}
__device__ __forceinline__ float dpp_from_next(float v) {      // lane j gets lane j + 1 (lane 63 gets lane 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x134, 0xf, 0xf, false));   // wave_rol:1
}

// The memory operations of the step loop are inline asm: exactly three per step and lane (two loads, one store), in a
// fixed order, so that "the loads issued PF steps ago have landed" is ONE counted wait (s_waitcnt vmcnt(3 PF - 2): that many
// younger operations may still be in flight) -- the compiler's own bookkeeping gives up at vmcnt(0) once stores sit
// between a load and its use, which would expose a memory round trip per step.
// a pointer that is wave-uniform by construction, as SGPRs whatever the register allocator made of it ("s" operands below).
// The v_readfirstlane it may cost writes SGPRs that the very next instruction -- inside the asm, invisible to the
// compiler's hazard recogniser -- reads as an address: "VALU writes SGPR -> VMEM reads it" needs five wait states, hence
// the `s_nop 4` in front of every memory instruction below (without it the instruction used the PREVIOUS pointer).
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void row_load8(bool DEV, float2& dst, const float2* row, unsigned lane_off) {      // DEV: wave-uniform
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

// the value c in [lo, hi] (hi - lo < 6) that a lane's local time tau can take in copy I of the unrolled body, or NONE:
// tau = t - 6 m and t = T0 + 12 body + I, so tau = I (mod 6) in copy I -- one candidate per edge rule and copy, which makes
// every edge rule "compare tau with a constant, then move between two registers known at compile time"
constexpr int cand(int I, int lo, int hi) {
    for (int c = lo; c <= hi; ++c)
        if (((c - I) % 6 + 6) % 6 == 0) return c;
    return NONE;
}
constexpr int ring(int i) { return ((i % UNR) + UNR) % UNR; }

struct LaneState {
    float2 R[UNR], P[UNR], Lb[UNR];
    float Ab[UNR];
    float2 cnext;           // c(tau) of the coming step, requested from LDS a step ahead
    int tau, m;
};

struct StepCtx {
    const float2* Dg;       // this utterance's [rows][64] complex array
    const float* Ag;        // ... and its magnitudes
    float2* Dw;
    const float2* ctab;     // LDS: c(k), k mod 64
    int M, trash_row;
    float thr;
    bool past_only;
    unsigned lane8, lane4;
};

// STD: the weights of the reference's geometry as compile-time constants (lws_skew_weights.h) -- literal operands of the
// multiply-adds.  As kernel arguments the 66 values do not fit the scalar registers beside everything else: the compiler
// parked them in VGPR lanes and fetched every one with a v_readlane per tap (30 of ~340 instructions per step).
// DIN / DOUT: device-scope loads (the stage before sits in another workgroup) / stores (the stage after) -- properties of
// the wave (first / last of its workgroup), compiled in: as run-time flags they were two scalar branches per memory
// operation, and a wave that has its SIMD to itself pays an issue slot for every scalar instruction.
template <int I, bool STD, bool DIN, bool DOUT>
__device__ __forceinline__ void skew_step(LaneState& L, const StepCtx& C, const SkewConst& W, int n) {
#define WBU(p, c) (STD ? AVSI_LWS_STD_BU[p][c] : W.bu[p][c])
#define WBD(p, c) (STD ? AVSI_LWS_STD_BD[p][c] : W.bd[p][c])
#define WB0(p, c) (STD ? AVSI_LWS_STD_B0[p][c] : W.b0[p][c])
    const int t = T0 + n;                                  // wave-uniform
    // ---- the operands requested PF steps ago
    // (the wait takes the landing registers as operands: nothing may read them, or give them to another value, before it)
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(L.Lb[I].x), "+v"(L.Lb[I].y), "+v"(L.Ab[I]) : "n"(3 * PF - 2) : "memory");
    const float2 arr = L.Lb[I];
    const float amp = L.Ab[I];
    // (addresses: the utterance's base pointer in SGPRs for the whole kernel + a 32-bit byte offset per lane that includes the
    // row -- one v_add per access instead of 64-bit pointer arithmetic and a v_readfirstlane pair per row)
    row_load8(DIN, L.Lb[ring(I + PF)], C.Dg, C.lane8 + (unsigned)(t + PF + 16 > 0 ? t + PF + 16 : 0) * (unsigned)(LANES * sizeof(float2)));     // position tau + 11 of step n + PF
    row_load4(L.Ab[ring(I + PF)], C.Ag, C.lane4 + (unsigned)(t + PF + ROW_OFF > 0 ? t + PF + ROW_OFF : 0) * (unsigned)(LANES * sizeof(float)));   // magnitude of the bin of step n + PF
    const int tau = L.tau, m = L.m;
    L.R[I] = arr;                                                                      // position tau + 11
    {   // positions 1 .. 5 of a row also define its mirror images -1 .. -5, which "arrived" 2 x steps earlier
        constexpr int x = cand(I + 11, 1, 5);
        if (x != NONE) L.R[ring(I - 2 * x)] = sel(tau + 11 == x, conjf2(arr), L.R[ring(I - 2 * x)]);
    }
    // ---- three sums over this lane's own rings (newest elements last: they end the dependence chain)
    // (the first tap of a sum: two products and one multiply-add -- a literal weight where `fma(w, x, 0)` would need an SGPR)
#define FIRST(w0, w1, val) make_float2(fmaf(w0, (val).x, -(w1) * (val).y), fmaf(w0, (val).y, (w1) * (val).x))
    float2 up = FIRST(WBU(0, 0), WBU(0, 1), L.R[ring(I - 2 * LMAX)]);
#pragma unroll
    for (int p = -LMAX + 1; p <= LMAX; ++p) up = cmadd(up, WBU(p + LMAX, 0), WBU(p + LMAX, 1), L.R[ring(I - LMAX + p)]);
    float2 own = FIRST(WB0(LMAX + 1, 0), WB0(LMAX + 1, 1), L.R[ring(I - 10)]);
#pragma unroll
    for (int p = 2; p <= LMAX; ++p) own = cmadd(own, WB0(LMAX + p, 0), WB0(LMAX + p, 1), L.R[ring(I - 11 + p)]);
    float2 dn = FIRST(WBD(0, 0), WBD(0, 1), L.P[ring(I - (SKEW + LMAX))]);
#pragma unroll
    for (int p = -LMAX + 1; p <= LMAX; ++p) dn = cmadd(dn, WBD(p + LMAX, 0), WBD(p + LMAX, 1), L.P[ring(I - (SKEW - p))]);
#undef FIRST
#pragma unroll
    for (int p = LMAX; p >= 1; --p) own = cmadd(own, WB0(LMAX - p, 0), WB0(LMAX - p, 1), L.P[ring(I - p)]);
    // ---- exchange: the row below (old values) from lane j + 1, the row above (new values) from lane j - 1
    float2 upr = make_float2(dpp_from_next(up.x), dpp_from_next(up.y));
    float2 dnr = make_float2(dpp_from_prev(dn.x), dpp_from_prev(dn.y));
    const float2 c = L.cnext;
    L.cnext = C.ctab[(tau + 1) & 63];          // (a wrap to the next frame moves tau by 384 = 6 x 64: the same entry)
    // (no select "zero where the neighbour frame does not exist": a lane with a frame >= M holds nothing but zeros -- its cells
    // are never written, what arrives is zero, what is handed on is what arrived -- and so does lane 63 before its first frame
    // has begun, which is frame -1 to lane 0)
    if (C.past_only) own = make_float2(0.f, 0.f), upr = own;
    float2 T = own;
    T.x = fmaf(c.x, upr.x, fmaf(-c.y, upr.y, T.x)), T.y = fmaf(c.x, upr.y, fmaf(c.y, upr.x, T.y));          // c . up
    T.x = fmaf(c.x, dnr.x, fmaf(c.y, dnr.y, T.x)), T.y = fmaf(c.x, dnr.y, fmaf(-c.y, dnr.x, T.y));          // conj(c) . down
    const float n2 = fmaf(T.x, T.x, T.y * T.y);          // (written as the fused form: the compiler's own choice differs from copy to copy)
    const bool valid = tau >= 0 && tau <= KB - 1 && m < C.M;
    const bool upd = valid && amp > C.thr && n2 > 0.f;
    const float sc = amp * __builtin_amdgcn_rsqf(n2);
    const float2 old = L.R[ring(I - 11)];                                              // position tau
    const float2 v = sel(upd, make_float2(T.x * sc, T.y * sc), old);
    // ---- what this lane hands on for position tau: the bin, or a mirror image
    float2 out = v;
    {   // below DC (tau in [-5, -1]): conj of the old bins 5 .. 1, the start values of the "new" ring
        constexpr int tc = cand(I, -LMAX, -1);
        if (tc != NONE) out = sel(tau == tc, conjf2(L.R[ring(I - (11 + 2 * tc))]), out);
    }
    {   // above Nyquist (tau in [257, 261]): conj of the NEW bins 255 .. 251
        constexpr int tc = cand(I, KB, TAU_LAST);
        if (tc != NONE) out = sel(tau == tc, conjf2(L.P[ring(I - (2 * tc - 2 * (KB - 1)))]), out);
    }
    L.P[I] = out;
    {   // bins 1 .. 5 refresh their mirror images below DC (2 x steps back in the "new" ring) ...
        constexpr int x = cand(I, 1, LMAX);
        if (x != NONE) L.P[ring(I - 2 * x)] = sel(tau == x, conjf2(v), L.P[ring(I - 2 * x)]);
    }
    {   // ... and bins 251 .. 255 theirs above Nyquist (positions 512 - tau of the "old" ring, still ahead of this lane)
        constexpr int tc = cand(I, KB - 1 - LMAX, KB - 2);
        if (tc != NONE) {
            constexpr int d = 11 - (2 * (KB - 1) - 2 * tc);            // slots back from the newest arrival
            L.R[ring(I - d)] = sel(tau == tc, conjf2(v), L.R[ring(I - d)]);
        }
    }
    // ---- store (row t + 5, every lane, always: lanes with nothing to store write a scratch row)
    const bool st = tau >= 0 && tau <= TAU_LAST && m < C.M;
    const unsigned off = C.lane8 + (unsigned)(st ? t + ROW_OFF : C.trash_row) * (unsigned)(LANES * sizeof(float2));
    row_store8(DOUT, C.Dw, off, out);
    // ---- next step of this lane
    const int nt = tau + 1;
    const bool wrap = nt > TAU_LAST;
    L.tau = wrap ? nt - PERIOD : nt;
    L.m = wrap ? m + LANES : m;
    __builtin_amdgcn_sched_barrier(0);          // steps are not interleaved: their live ranges would add up
#undef WBU
#undef WBD
#undef WB0
}

template <bool STD, int H, bool DIN, bool DOUT, int... Is>
__device__ __forceinline__ void skew_half(LaneState& L, const StepCtx& C, const SkewConst& W, int n0, std::integer_sequence<int, Is...>) {
    (skew_step<H * HALF + Is, STD, DIN, DOUT>(L, C, W, n0 + H * HALF + Is), ...);
}

struct StageCtx {           // where a wave sits in the chain of its utterance, and the counters of that chain
    int wv, wg, lane, stage, stages, nb, hlast, nbp;
    int* gprog;
    volatile int* vprog;
    int* status;
    float mean, amax;
};

// the sweeps of one stage.  DIN / DOUT (first / last wave of a workgroup) are compiled in: the kernel enters one of three
// copies of this function once, so the step loop has no branch on the wave's role and each copy its own register allocation
template <int NW, bool STD, bool DIN, bool DOUT>
__device__ __forceinline__ void skew_sweeps(StepCtx& C, const SkewConst& W, const AvsiLwsSchedule& sched, const StageCtx& Q) {
    const int wv = Q.wv, wg = Q.wg, lane = Q.lane, stage = Q.stage, stages = Q.stages, nb = Q.nb, hlast = Q.hlast, nbp = Q.nbp;
    int* gprog = Q.gprog;
    volatile int* vprog = Q.vprog;
    int* status = Q.status;
    const float mean = Q.mean, amax = Q.amax;
    constexpr bool dev_in = DIN, dev_out = DOUT;
    bool dead = false;
    int known = 0;            // last value read from the predecessor's counter
    int mine = 0;             // sweeps this stage has finished

    int rank_a = -1, last_active = -1;
    for (int sw = 0; sw < sched.n; ++sw) {
        const float thr = sched.rel[sw] * mean;
        if (!(amax > thr)) continue;              // an idle sweep changes nothing and needs no stage
        const int pred = last_active;
        last_active = sw, ++rank_a;
        if (rank_a % stages != stage) continue;
        C.thr = thr;
        C.past_only = sched.past_only[sw] != 0;
        const int pstage = pred < 0 ? -1 : (rank_a - 1) % stages;
        const int pbase = pred < 0 ? 0 : ((rank_a - 1) / stages) * nbp;       // counter value at the start of the predecessor's sweep
        const int base = mine * nbp;
        LaneState L;
#pragma unroll
        for (int i = 0; i < UNR; ++i) L.R[i] = L.P[i] = L.Lb[i] = make_float2(0.f, 0.f), L.Ab[i] = 0.f;
        L.m = lane;
        L.tau = T0 - SKEW * lane;
        L.cnext = C.ctab[L.tau & 63];
        // a stage whose predecessor sits in another workgroup polls device memory, which drains the loads in flight: it
        // asks for QSLACK half bodies more than it needs and then runs that many on what it knows
        auto wait_pred = [&](int half) {           // half: index of the half body about to start
            if (pstage < 0 || dead) return;
            const int need = pbase + (half + AHEAD < hlast ? half + AHEAD : hlast);
            if (known >= need) return;
            const int want = dev_in ? (need + QSLACK < pbase + hlast ? need + QSLACK : pbase + hlast) : need;
            int spins = 0;
            for (;;) {
                known = dev_in ? __hip_atomic_load(gprog + (pstage / NW) * CTR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : vprog[wv - 1];
                known = __builtin_amdgcn_readfirstlane(known);
                if (known >= want) break;
                // a predecessor that is still bodies away is not worth asking often: with hundreds of stages waiting for their
                // turn, their polls alone kept the memory side busy (32 utterances x 26 workgroups: 20 ms instead of 5)
                if (known + 4 < want) __builtin_amdgcn_s_sleep(127);
                else __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 22)) {
                    dead = true;
                    if (lane == 0 && status) atomicOr(status, 1);
                    break;
                }
            }
        };
        auto publish = [&](int value) {
            if (lane == 0) {
                vprog[wv] = value;
                if (dev_out) __hip_atomic_store(gprog + wg * CTR_STRIDE, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        // prefetch: the rows of steps 0 .. PF - 1 (nothing of them is used before position 1 arrives)
        wait_pred(0);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            row_load8(dev_in, L.Lb[i], C.Dg, C.lane8 + (unsigned)(T0 + i + 16 > 0 ? T0 + i + 16 : 0) * (unsigned)(LANES * sizeof(float2)));     // (rows < 0: positions nobody reads)
            row_load4(L.Ab[i], C.Ag, C.lane4 + (unsigned)(T0 + i + ROW_OFF > 0 ? T0 + i + ROW_OFF : 0) * (unsigned)(LANES * sizeof(float)));
            // keep the count of the step loop: a store per step (scratch row)
            row_store8(false, C.Dw, C.lane8 + (unsigned)C.trash_row * (unsigned)(LANES * sizeof(float2)), make_float2(0.f, 0.f));
        }
        constexpr std::make_integer_sequence<int, HALF> seq{};
        for (int body = 0; body < nb; ++body) {
            const int n0 = body * UNR;
            wait_pred(2 * body);
            skew_half<STD, 0, DIN, DOUT>(L, C, W, n0, seq);
            // everything issued before this half body's 18 operations is complete: the stores of the half bodies before it
            asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            publish(base + 2 * body);
            wait_pred(2 * body + 1);
            skew_half<STD, 1, DIN, DOUT>(L, C, W, n0, seq);
            asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            publish(base + 2 * body + 1);
        }
        // the loads of the last body land in registers nobody reads any more: they stay reserved until they have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < UNR; ++i) asm volatile("" : "+v"(L.Lb[i].x), "+v"(L.Lb[i].y), "+v"(L.Ab[i]));
        publish(base + hlast);
        ++mine;
    }
}

// NW waves per workgroup = NW pipeline stages of one utterance; G workgroups per utterance
// (sixteen waves per CU whatever NW: the host counts on that many resident workgroups -- four waves per SIMD = 128 VGPRs)
template <int NW, bool STD>
__global__ __launch_bounds__(64 * NW, 4) void lws_skew_kernel(float2* __restrict__ Dall, const float* __restrict__ Aall, int B, int M,
                                                          const SkewConst W, const AvsiLwsSchedule sched, int* __restrict__ status,
                                                          const float2* __restrict__ stats, int* __restrict__ gprog_all, int G) {
    __shared__ float2 ctab[64];
    __shared__ int prog[NW];                    // bodies finished and visible, per stage of this workgroup (monotone)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // wv: wave-uniform, and known to be
    const int b = blockIdx.x / G, wg = blockIdx.x - b * G;
    const int stage = wg * NW + wv, stages = G * NW;
    const int rows = skew_rows(M);
    const int nb = (skew_steps(M) + UNR - 1) / UNR;           // bodies per sweep
    const int hlast = 2 * nb + AHEAD + 1;                     // published when a sweep is finished: every need is capped there
    const int nbp = 2 * nb + 16;                              // counter values per sweep of a stage
    // one counter per workgroup of the chain (its last stage publishes there), each on a 256-byte line of its own: polls
    // and publications are device-scope accesses served at the memory side, and neighbours in one line queue on one channel
    int* gprog = gprog_all + (size_t)b * GPROG_INTS;
    volatile int* vprog = prog;
    if (threadIdx.x < 64) {
        float sn, cs;
        sincospif(-2.f * (float)((threadIdx.x * W.phase_step) & 63) / 64.f, &sn, &cs);
        ctab[threadIdx.x] = make_float2(cs, sn);
    }
    if (threadIdx.x < NW) prog[threadIdx.x] = 0;
    __syncthreads();
    const float2 st = stats[b];
    const float mean = st.x, amax = st.y;
    StepCtx C;
    C.Dg = uniform_ptr(Dall + (size_t)b * rows * LANES);        // SGPRs from here on: the "s" operands of the memory instructions
    C.Dw = uniform_ptr(Dall + (size_t)b * rows * LANES);
    C.Ag = uniform_ptr(Aall + (size_t)b * rows * LANES);
    C.ctab = ctab;
    C.M = M;
    C.trash_row = rows - 1;
    C.lane8 = lane * 8u, C.lane4 = lane * 4u;
    StageCtx Q;
    Q.wv = wv, Q.wg = wg, Q.lane = lane, Q.stage = stage, Q.stages = stages, Q.nb = nb, Q.hlast = hlast, Q.nbp = nbp;
    Q.gprog = gprog, Q.vprog = vprog, Q.status = status, Q.mean = mean, Q.amax = amax;
    // the stage before the first wave / after the last one sits in another workgroup (or is this one in the next round)
    if (wv == 0) skew_sweeps<NW, STD, true, false>(C, W, sched, Q);
    else if (wv == NW - 1) skew_sweeps<NW, STD, false, true>(C, W, sched, Q);
    else skew_sweeps<NW, STD, false, false>(C, W, sched, Q);
}

// spec [B][M][257] -> diagonal layout (bins, the five mirror positions above Nyquist, magnitudes); everything else zero
__global__ __launch_bounds__(256) void lws_to_diag_kernel(const float2* __restrict__ spec, int M, int rows, float2* __restrict__ D,
                                                         float* __restrict__ A) {
    const int b = blockIdx.y, l = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int x0 = r - ROW_OFF - SKEW * l;                    // position if the frame were l
    float2 v = make_float2(0.f, 0.f);
    float a = 0.f;
    if (x0 >= 0) {
        const int rnd = x0 / PERIOD, x = x0 - rnd * PERIOD, m = rnd * LANES + l;
        if (x <= TAU_LAST && m < M) {
            const float2* sp = spec + ((int64_t)b * M + m) * KB;
            if (x < KB) {
                v = sp[x];
                a = sqrtf(fmaf(v.x, v.x, v.y * v.y));      // (the fused form, spelled out: lws_duo.hip takes the same)
            } else {
                v = conjf2(sp[2 * (KB - 1) - x]);
            }
        }
    }
    D[((int64_t)b * rows + r) * LANES + l] = v;
    A[((int64_t)b * rows + r) * LANES + l] = a;
}

__global__ __launch_bounds__(256) void lws_from_diag_kernel(const float2* __restrict__ D, int M, int rows, float2* __restrict__ spec) {
    const int b = blockIdx.y, l = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int x0 = r - ROW_OFF - SKEW * l;
    if (x0 < 0) return;
    const int rnd = x0 / PERIOD, x = x0 - rnd * PERIOD, m = rnd * LANES + l;
    if (x < KB && m < M) spec[((int64_t)b * M + m) * KB + x] = D[((int64_t)b * rows + r) * LANES + l];
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// steps (of one wave) until the last of `sweeps` sweeps of `len` steps has finished on `stages` pipeline stages
double skew_pipeline_steps(int sweeps, int stages, int len, int lag) {
    double start[MAX_SWEEPS + 1];
    for (int a = 0; a < sweeps; ++a) {
        const double chain = a ? start[a - 1] + lag : 0.0, stage = a >= stages ? start[a - stages] + len : 0.0;
        start[a] = chain > stage ? chain : stage;
    }
    return start[sweeps - 1] + len;
}

// The launch shape for a batch.  NW by rule (up to 32 utterances 4, up to 128: 8, up to 256: 8 or 16, beyond: 16): the more
// stages share a workgroup, the fewer hand-overs go through device memory -- at 1024 utterances 16 x 1 takes 57.5 ms,
// 8 x 1 72, 4 x 1 95 -- but a wave that shares its SIMD with three others is slow, so small batches take 4-wave
// workgroups.  G (and NW where the rule leaves two) under a model of the pipeline: sweep a starts LAG steps after
// sweep a - 1 and not before its stage has finished sweep a - stages; a step costs a wave 0.6 us with a SIMD to itself
// (non-VALU instructions and dependent issue: 55 % of the VALU rate), 0.75 with two, 1.28 with four waves on the SIMD of
// the most loaded CU (which the batch waits for).  Measured at a lag of 36 steps, ms per batch: 1 .. 8 utterances 4 x 13: 3.7; 32: 4 x 8 4.2
// (4 x 12: 4.6, 4 x 16: 4.7); 64: 8 x 4 5.5 (8 x 6: 6.9, 8 x 8: 7.2); 100: 8 x 5 7.7 (8 x 3: 10.1); 128: 8 x 2 8.8 (8 x 4: 9.3);
// 160: 8 x 3 10.7 (16 x 1: 12.7, 16 x 3 in two launches: 16.9); 200: 16 x 1 12.9 (8 x 2: 14.8).
// A given nw / g (non-zero) is kept.
void skew_shape(int batch, int sweeps, int len, int& nw, int& g) {
    const int lag = (AHEAD + 1) * HALF;
    if (sweeps > MAX_SWEEPS) sweeps = MAX_SWEEPS;
    double best_t = 0.0;
    int best_nw = 0, best_g = 0;
    for (int cnw : {4, 8, 16}) {
        if (nw ? cnw != nw : (batch <= 32 ? cnw != 4 : (batch <= 128 ? cnw != 8 : (batch <= 256 ? cnw == 4 : cnw != 16)))) continue;
        const int capacity = AVSI_NUM_CU * (16 / cnw);
        double cost[MAX_SWEEPS / 4 + 1], best = 0.0;
        int n = 0;
        for (int cg = 1; cg <= MAX_SWEEPS / 4; ++cg) {
            if (cg > capacity || cg * cnw > MAX_SWEEPS || (cg > 1 && (cg - 1) * cnw >= sweeps)) break;
            const int per_launch = capacity / cg, launches = (batch + per_launch - 1) / per_launch;
            const int resident = batch < per_launch ? batch : per_launch;
            const int wg_per_cu = (resident * cg + AVSI_NUM_CU - 1) / AVSI_NUM_CU;
            const double w = wg_per_cu * cnw / 4.0;                  // waves per SIMD on the most loaded CU
            // (the waves of a SIMD get on best when they are consecutive stages of one utterance: one 16-wave workgroup
            // 1.05 us per step at w = 4, two 8-wave ones 1.2; two 4-wave ones 0.85 at w = 2, one 8-wave one 0.75)
            const double step_us = w <= 1.0 ? 0.6 : 0.30 * w + 0.30 / w + (cnw == 4 ? 0.1 : (cnw == 16 ? -0.2 : 0.0));
            // (chained 16-wave workgroups: 129 utterances 16 x 3 took 16.9 ms where this model without the factor says 12)
            cost[cg] = launches * step_us * skew_pipeline_steps(sweeps, cg * cnw, len, lag) * (cnw == 16 && cg > 1 ? 1.4 : 1.0);
            if (n == 0 || cost[cg] < best) best = cost[cg];
            n = cg;
        }
        int pick = 0;
        for (int cg = 1; cg <= n && pick == 0; ++cg)
            if (g ? cg == g : cost[cg] <= 1.05 * best) pick = cg;    // within 5 % of the cheapest: the smallest shape
        if (pick && (best_nw == 0 || cost[pick] < best_t)) best_t = cost[pick], best_nw = cnw, best_g = pick;
    }
    if (best_nw) nw = best_nw, g = best_g;
    else nw = nw ? nw : 8, g = g ? g : 1;         // (a given shape outside the table: the caller's range checks decide)
}

}  // namespace

// word 0: status; (mean, max) per utterance; one row of stage counters per utterance; the diagonal arrays
extern "C" size_t avsi_lws_run_skew_workspace_bytes(int batch, int num_frames) {
    if (batch <= 0 || num_frames <= 0) return 0;
    const size_t cells = (size_t)batch * skew_rows(num_frames) * LANES;
    return align256(16 + (size_t)batch * sizeof(float2)) + align256((size_t)batch * GPROG_INTS * sizeof(int)) +
           align256(cells * sizeof(float2)) + align256(cells * sizeof(float));
}

// the launch shape avsi_lws_run_skew_f32 takes for a batch (a given, non-zero *waves_per_group / *groups_per_utterance is kept)
extern "C" int avsi_lws_skew_launch_shape(int batch, int num_frames, int sweeps, int* waves_per_group, int* groups_per_utterance) {
    if (batch <= 0 || num_frames <= 0 || sweeps <= 0 || !waves_per_group || !groups_per_utterance) return AVSI_ERR_INVALID_ARG;
    if (*waves_per_group != 0 && *waves_per_group != 4 && *waves_per_group != 8 && *waves_per_group != 16) return AVSI_ERR_INVALID_ARG;
    skew_shape(batch, sweeps, skew_steps(num_frames), *waves_per_group, *groups_per_utterance);
    return AVSI_OK;
}

extern "C" int avsi_lws_run_skew_f32(float* spec, int batch, int num_frames, int frame_len, int hop, int nfft, int L,
                                     int nofuture_iterations, float nofuture_alpha, int online_iterations, float online_alpha,
                                     int batch_iterations, float batch_alpha, float batch_beta, float batch_gamma,
                                     int waves_per_group, int groups_per_utterance, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!spec || batch <= 0 || num_frames <= 0 || L < 1 || nofuture_iterations < 0 || online_iterations < 0 ||
        batch_iterations < 0)
        return AVSI_ERR_INVALID_ARG;
    AvsiLwsSchedule S;
    if (!avsi_lws_geometry_ok(frame_len, hop, nfft) || L > LMAX ||
        !avsi_lws_make_schedule(nofuture_iterations, nofuture_alpha, online_iterations, online_alpha, batch_iterations,
                                batch_alpha, batch_beta, batch_gamma, S))
        return AVSI_ERR_UNSUPPORTED;
    if (S.n == 0) return AVSI_OK;
    if (!workspace || workspace_bytes < avsi_lws_run_skew_workspace_bytes(batch, num_frames)) return AVSI_ERR_WORKSPACE;
    // b_q(p) = alpha_q(p) e^{-2 pi j p q R / N}
    double alpha[3][NP][2];
    avsi_lws_host_alpha(frame_len, hop, nfft, L, alpha);
    SkewConst W;
    for (int p = -LMAX; p <= LMAX; ++p) {
        const double ph = -2.0 * M_PI * p * hop / nfft, cs = cos(ph), sn = sin(ph);
        const double *au = alpha[2][p + LMAX], *ad = alpha[0][p + LMAX], *a0 = alpha[1][p + LMAX];
        W.bu[p + LMAX][0] = (float)(au[0] * cs - au[1] * sn), W.bu[p + LMAX][1] = (float)(au[0] * sn + au[1] * cs);
        W.bd[p + LMAX][0] = (float)(ad[0] * cs + ad[1] * sn), W.bd[p + LMAX][1] = (float)(-ad[0] * sn + ad[1] * cs);
        W.b0[p + LMAX][0] = (float)a0[0], W.b0[p + LMAX][1] = (float)a0[1];
    }
    W.phase_step = 64 * hop / nfft;
    // the reference's geometry: the same weights as compile-time constants (checked here against the ones just computed)
    bool standard = frame_len == AVSI_LWS_STD_FRAME && hop == AVSI_LWS_STD_HOP && nfft == AVSI_LWS_STD_NFFT && L == AVSI_LWS_STD_L;
    for (int p = 0; p < NP && standard; ++p)
        for (int c = 0; c < 2; ++c)
            standard = fabs((double)W.bu[p][c] - AVSI_LWS_STD_BU[p][c]) <= 1e-7 && fabs((double)W.bd[p][c] - AVSI_LWS_STD_BD[p][c]) <= 1e-7 &&
                       fabs((double)W.b0[p][c] - AVSI_LWS_STD_B0[p][c]) <= 1e-7;
    if (frame_len == AVSI_LWS_STD_FRAME && hop == AVSI_LWS_STD_HOP && nfft == AVSI_LWS_STD_NFFT && L == AVSI_LWS_STD_L && !standard)
        return AVSI_ERR_UNSUPPORTED;          // lws_skew_weights.h is stale: regenerate it (tools/gen_lws_skew_weights.py)
    // Launch shape: NW stages (waves) per workgroup, G workgroups chained per utterance, all of them resident together (a CU
    // holds 16 waves of this kernel: 128 VGPRs) -- see skew_shape().
    if (waves_per_group != 0 && waves_per_group != 4 && waves_per_group != 8 && waves_per_group != 16) return AVSI_ERR_INVALID_ARG;
    // (each launch of a batch that needs several takes the shape of what is left: 257 utterances = 256 on 16 x 1, one on 4 x 12)
    auto shape_for = [&](int utterances, int& nw, int& g) {
        nw = standard ? waves_per_group : 8, g = groups_per_utterance;
        if (nw == 0 || g == 0) skew_shape(utterances, S.n, skew_steps(num_frames), nw, g);
        const int cap = AVSI_NUM_CU * (16 / nw);              // workgroups the chip holds at once
        return g >= 1 && g * nw <= MAX_SWEEPS && g <= MAX_SWEEPS / 4 && g <= cap;
    };
    int NW, G;
    if (!shape_for(batch, NW, G)) return AVSI_ERR_INVALID_ARG;
    const hipStream_t st = (hipStream_t)stream;
    const int rows = skew_rows(num_frames);
    char* ws = static_cast<char*>(workspace);
    int* status = reinterpret_cast<int*>(ws);
    float2* stats = reinterpret_cast<float2*>(ws + 16);
    size_t off = align256(16 + (size_t)batch * sizeof(float2));
    int* gprog = reinterpret_cast<int*>(ws + off);
    off += align256((size_t)batch * GPROG_INTS * sizeof(int));
    float2* D = reinterpret_cast<float2*>(ws + off);
    off += align256((size_t)batch * rows * LANES * sizeof(float2));
    float* A = reinterpret_cast<float*>(ws + off);
    if (hipMemsetAsync(workspace, 0, align256(16 + (size_t)batch * sizeof(float2)) + align256((size_t)batch * GPROG_INTS * sizeof(int)),
                       st) != hipSuccess)
        return AVSI_ERR_LAUNCH;
    avsi_clear_error();
    avsi_lws_launch_stats(spec, batch, num_frames, reinterpret_cast<float*>(stats), st);
    // every workgroup of a launch must be resident (its stages wait for each other): batches beyond the chip's capacity
    // run as consecutive launches
    for (int b0 = 0, nbatch = 0; b0 < batch; b0 += nbatch) {
        if (!shape_for(batch - b0, NW, G)) return AVSI_ERR_INVALID_ARG;
        const int per_launch = AVSI_NUM_CU * (16 / NW) / G;
        nbatch = batch - b0 < per_launch ? batch - b0 : per_launch;
        float2* sp = reinterpret_cast<float2*>(spec) + (size_t)b0 * num_frames * KB;
        float2* Db = D + (size_t)b0 * rows * LANES;
        float* Ab = A + (size_t)b0 * rows * LANES;
        hipLaunchKernelGGL(lws_to_diag_kernel, dim3((rows + 3) / 4, nbatch), dim3(256), 0, st, sp, num_frames, rows, Db, Ab);
#define AVSI_SKEW_LAUNCH(NWV, STDV)                                                                                          \
    hipLaunchKernelGGL((lws_skew_kernel<NWV, STDV>), dim3(nbatch* G), dim3(64 * (NWV)), 0, st, Db, Ab, nbatch, num_frames, W, S,   \
                       status, stats + b0, gprog + (size_t)b0 * GPROG_INTS, G)
        if (!standard) AVSI_SKEW_LAUNCH(8, false);            // other geometries: one shape, weights as kernel arguments
        else if (NW == 16) AVSI_SKEW_LAUNCH(16, true);
        else if (NW == 8) AVSI_SKEW_LAUNCH(8, true);
        else AVSI_SKEW_LAUNCH(4, true);
#undef AVSI_SKEW_LAUNCH
        hipLaunchKernelGGL(lws_from_diag_kernel, dim3((rows + 3) / 4, nbatch), dim3(256), 0, st, Db, num_frames, rows, sp);
    }
    return avsi_launch_status();
}
