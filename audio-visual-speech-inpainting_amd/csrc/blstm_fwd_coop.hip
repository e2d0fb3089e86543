// Small-batch form of the recurrent half of one bidirectional LSTM layer (same maths, layouts and
// packed operands as blstm_fwd.hip; reference models.py:95-115).
//
// blstm_fwd.hip is batch-stationary: one workgroup owns 32 / 64 utterances of a direction and
// streams the 1 MiB recurrent kernel from L2 every step.  At the reference's own batch sizes
// (8 for training, 32 for inference: blstm.config:8, inference.sh:7) that is two workgroups on a
// 256-CU chip and 37 us per time step, all of it latency: 512 dependent-free MFMAs per wave plus a
// 128 KiB fragment stream per wave and step.  Here a (32-utterance tile, direction) pair is spread
// over S = 4 or 8 workgroups ("weight-stationary"):
//   - workgroup m of the group owns 8 / S of the eight 32-unit slices of the hidden state (all four
//     gates of those units); inside it the reduction over h_{t-1} is split over S waves per slice.
//     Every wave keeps ITS piece of Wh (32 / S k-groups x 4 gates = 64 or 128 VGPRs) in registers
//     for all T steps: after the first step no weight is ever loaded again;
//   - per step a wave issues 64 (S = 8) or 128 (S = 4) MFMAs on h_{t-1} fragments read straight
//     from global memory (the previous step's hout rows, L2-resident), parks its partial
//     32 x 128 tile in LDS, and after one workgroup barrier the 512 lanes finish 2 (4) cells each:
//     sum of the S partials + hoisted input projection, LSTM cell, c_t in registers;
//   - the S workgroups exchange h_t through hout itself: device-coherent (sc1) stores, one atomic
//     increment of the group's step counter once they have completed, and the consumers spin on
//     that counter before they read hout[t] with device-coherent loads.  Members of a group are
//     placed S blocks of 8 apart (round-robin dispatch then tends to put them on one XCD), but that
//     is a locality hint only: nothing is assumed about placement -- an experiment with L2-scope
//     (sc0) atomics between the members never saw its peers' increments, so every exchange uses
//     device scope.  The spin is bounded: a group that does not see its peers within AVSI_COOP_TIMEOUT_MS (2 s of wall clock) marks
//     the status word, stops waiting and runs to the end, so the grid always drains.
// The launch needs all S members of a group resident together; groups are contiguous windows of
// 8 S block ids and the kernel uses one workgroup per CU, so in-order dispatch guarantees that for
// any grid (and the host only selects this kernel when the whole grid fits the chip anyway).
#include <stdlib.h>

#include "avsi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HP = 256, GP = 4 * HP;
constexpr int PSTRIDE = 36;                          // LDS pitch of one unit's 32 rows (+4: conflict-free b128)
constexpr int PART_FLOATS = 8 * 4 * 32 * PSTRIDE;    // [wave][gate][unit][row]
// One 256-byte line per step counter.  Packed counters (16 to a line) made every poller and every increment of
// 16 groups queue on one memory channel: at 512 workgroups that, not the exchange itself, set the step time
// (column-split kernel, 1024 utterances: 5.9 -> 3.0 ms per layer from this alone).
constexpr int CTR_STRIDE = 64;

struct CoopArgs {
    const float* xproj;
    const float* whp;
    float* hout;
    float* resv;
    unsigned* sync;    // [0] status, [CTR_STRIDE * (1 + group)] step counters
    int T, Bp, ngroups;
    int tile0;         // first 32-utterance tile of this launch (large batches run in resident-sized chunks)
    float* xch;        // fine kernels: h in exchange layout [T][group][member][32 utterances][units of the member], or null
    int xtile0, xtiles;   // ... of the tiles [xtile0, xtile0 + xtiles) of the CALL (a row-range call covers part of the batch)
    int coherent;      // AVSI_COOP_COHERENT=1: every load of exchanged bytes at device scope, whatever the invariants allow
    long long spin_ticks;   // bound of every wait for a peer, 100 MHz ticks (avsi_coop_spin_ticks)
    unsigned long long* stamps;   // diagnostics (STAMPS, avsi_diag_cs_stamps): [block < 32][step 64 .. 71][wave 0 / 1][8 phases]
    int noack;         // fine kernels with xch: publish WITHOUT waiting for the acknowledgement of the h store; the exchange copy
                       // was filled with COOP_POISON words before the launch and a reader that meets one loads again (see below)
};

// Poisoned exchange (AVSI_COOP_NOACK, round 5).  The per-step protocol store -> acknowledgement -> counter -> poll -> load pays
// the acknowledgement (0.2 - 0.3 us of a 2.7 us step at 32 utterances, tools/rec_fine_stamps.py) only so that the counter cannot
// overtake the data.  With every word of the launch's exchange copy preset to a bit pattern no h can have (all ones: a NaN
// the arithmetic never produces -- its NaNs are the canonical 0x7fc00000 or carry their operand's payload), the counter MAY
// overtake: a reader checks the 16 words it loaded and, if one is still the preset, loads again at device scope (an
// incomplete line may sit in its L1 / L2 by then) until the data is there, bounded like every other wait.  Each word is
// written once per launch, so "not the preset" means "final".
constexpr unsigned COOP_POISON = 0xFFFFFFFFu;

__device__ __forceinline__ float sigmoidf_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_fast(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

// 16-byte loads that are coherent at device scope (sc0 sc1: served at the memory side, never from
// this CU's L1 or a stale line of the XCD's L2).  Inline asm because the only portable spelling --
// relaxed agent-scope atomic loads -- exists per dword (4 x the instructions and 4 x the address
// VALU work; the BPTT kernel reads 16 fragments per lane and step).  The compiler does not know
// these loads are in flight, so a batch is closed by coherent_wait(): one s_waitcnt, after which
// every loaded value is passed through an empty asm so that no consumer can be scheduled before it.
typedef float v4f __attribute__((ext_vector_type(4)));
// byte_off: a constant after unrolling -- one address register pair serves a whole batch.
// PLAIN drops the scope bits: the load may then be served by this XCD's L2 (or the CU's L1).  That is correct only
// under an invariant of the kernel that uses it: within a launch every 128-byte line of the exchange buffer is written
// once, in full, before the step counter that releases its readers moves, NO load of any kind touches it earlier on
// any CU (so no cache can hold a copy from before it was complete), and the caches hold nothing of it from before the
// launch (kernel boundaries invalidate them).  The 4- and 8-way BPTT kernels satisfy it and read 64 - 128 KB of dz per
// workgroup and step this way (the members of a group sit on one XCD, so all but the first reader of a line hit in its
// L2: 2.13 -> 1.85 ms per layer at 512 utterances, 3.43 -> 3.16 at 1024).  The fine kernels do NOT: they touch lines
// ahead of time (see below), which leaves early copies in the toucher's L1 and L2, so every load of exchanged bytes
// there is a device-scope load (agent_load4_issue) -- the rule of cdna_hip_programming.md, Guideline 16.
template <bool PLAIN = false>
__device__ __forceinline__ void coherent_load4_issue(v4f& dst, const float* p, int byte_off) {
    if (PLAIN)
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "i"(byte_off) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0 sc1" : "=v"(dst) : "v"(p), "i"(byte_off) : "memory");
}
// (`sc1` alone, the agent-scope form, measured the same as `sc0 sc1`: 0.825 / 0.853 against 0.824 / 0.866 ms per layer at 32 /
// 128 utterances.)
__device__ __forceinline__ void agent_load4_issue(v4f& dst, const float* p, int byte_off) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0 sc1" : "=v"(dst) : "v"(p), "i"(byte_off) : "memory");
}
// After its last step, member 0 of a group waits until every member has published that step (nobody polls the
// counter any more) and puts the counter back to zero: the workspace is left as it was found, so the next launch on
// it needs no memset in front (a 5 us kernel and two launch gaps per recurrent launch -- 0.12 ms of a 7 ms training
// step at 32 utterances).  Not after a bounded wait has given up: the status word says so and the caller re-zeroes.
__device__ __forceinline__ void reset_counter_when_done(unsigned* ctr, unsigned total, long long spin_ticks, const unsigned* status) {
    unsigned polls = 0;
    long long t0 = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) {
        __builtin_amdgcn_s_sleep(2);
        if (avsi_spin_expired(polls, t0, spin_ticks, status)) return;
    }
    __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int N>
__device__ __forceinline__ void coherent_wait(v4f (&v)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

template <int S, bool SAVE>
__global__ __launch_bounds__(512, 2) void blstm_rec_fwd_coop_kernel(const CoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = reinterpret_cast<float*>(smem);
    __shared__ int dead;                              // this workgroup gave up waiting (status word set)

    constexpr int UG = 8 / S;                         // 32-unit slices per workgroup
    constexpr int QPW = 32 / S;                       // 8-wide k groups per wave
    constexpr int CPL = 2 * UG;                       // (row, unit) cells finished per lane

    // block -> (group, member): members of a group are 8 block ids apart (same XCD)
    const int xcd = blockIdx.x % AVSI_NUM_XCD, kk = blockIdx.x / AVSI_NUM_XCD;
    const int member = kk % S;
    const int group = (kk / S) * AVSI_NUM_XCD + xcd;
    if (group >= a.ngroups) return;
    // behind a launch that gave up a bounded wait every result is void (sticky status word): leave at once instead of
    // waiting out the time bound on step counters that launch may have left behind
    if (avsi_launch_is_void(a.sync)) return;
    const int dir = group & 1;
    const int b0 = (a.tile0 + (group >> 1)) * 32;
    const int T = a.T, Bp = a.Bp;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int ugl = v / S, ks = v % S;                // slice of this wave, its part of the reduction
    const int wq = member * UG + ugl;                 // slice index 0..7 in the packed layouts
    if (tid == 0) dead = 0;

    // ---- this wave's piece of Wh, once: whp [2][8 w][32 q][4 g][64 lane][4 s]
    float4 wreg[QPW][4];
    {
        const float4* wp = reinterpret_cast<const float4*>(a.whp) + (size_t)(dir * 8 + wq) * (32 * 4 * 64) + lane;
#pragma unroll
        for (int q = 0; q < QPW; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) wreg[q][g] = wp[((ks * QPW + q) * 4 + g) * 64];
    }

    // ---- finishing role of this lane: unit fu of slice (member * UG + c / 2), rows frow, frow + 1
    const int fu = tid & 31, frow = (tid >> 5) * 2;
    float cstate[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) cstate[c] = 0.f;

    unsigned* ctr = a.sync + CTR_STRIDE * (1 + 2 * a.tile0 + group);
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        const int tprev = dir ? t + 1 : t - 1;
        const size_t row0 = (size_t)t * Bp + b0;

        // hoisted input projection of this lane's cells: in flight during the wait and the MFMAs
        // (S = 8), or fetched while the partial tiles are parked (S = 4, whose weights leave no room)
        float xz[CPL][4];
        auto load_xz = [&]() {
#pragma unroll
            for (int c = 0; c < CPL; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    xz[c][g] = a.xproj[(row0 + frow + (c & 1)) * (2 * GP) + dir * GP + (member * UG + (c >> 1)) * 128 + g * 32 + fu];
        };
        if (S == 8) load_xz();

        f32x16 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

        if (step > 0) {
            // ---- wait until all S members have published h of the previous step
            if (tid == 0 && !dead) {
                const unsigned want = (unsigned)S * (unsigned)step;
                unsigned polls = 0;
                long long t0 = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (avsi_spin_expired(polls, t0, a.spin_ticks, a.sync)) {
                        dead = 1;
                        atomicExch(a.sync, 1u);
                        break;
                    }
                }
            }
            __syncthreads();

            // ---- partial z = h_{t-1}[:, k range of this wave] . Wh piece.  The h exchange uses
            //      device-coherent accesses (sc1: served at the memory side, not from a possibly stale
            //      L1 / other XCD's L2) instead of cache-wide acquire / release fences, whose cost
            //      (L2 write-back + invalidate per workgroup and step) grew with the number of groups.
            const float* hp = a.hout + ((size_t)tprev * Bp + b0 + li) * (2 * HP) + dir * HP + 4 * hi;
            // fragments in batches of 4 (the weights already hold 64 / 128 registers)
#pragma unroll
            for (int c0 = 0; c0 < QPW; c0 += 4) {
                v4f af[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) coherent_load4_issue(af[q], hp + 8 * ks * QPW, 32 * (c0 + q));
                coherent_wait(af);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 b = wreg[c0 + q][g];
                            const float av = s == 0 ? af[q].x : s == 1 ? af[q].y : s == 2 ? af[q].z : af[q].w;
                            const float bv = s == 0 ? b.x : s == 1 ? b.y : s == 2 ? b.z : b.w;
                            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[g], 0, 0, 0);
                        }
            }
        }

        // ---- park the partial tile: part[v][g][unit li][row], rows (r&3) + 8 (r>>2) + 4 hi
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(part + ((v * 4 + g) * 32 + li) * PSTRIDE + 8 * j + 4 * hi) =
                    make_float4(acc[g][4 * j], acc[g][4 * j + 1], acc[g][4 * j + 2], acc[g][4 * j + 3]);
        if (S != 8) load_xz();
        __syncthreads();

        // ---- finish this lane's cells
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int slice = c >> 1, r = frow + (c & 1);
            float z[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = xz[c][g];
#pragma unroll
                for (int k = 0; k < S; ++k) s += part[(((slice * S + k) * 4 + g) * 32 + fu) * PSTRIDE + r];
                z[g] = s;
            }
            const float ig = sigmoidf_fast(z[0]), jg = tanhf_fast(z[1]), fg = sigmoidf_fast(z[2]), og = sigmoidf_fast(z[3]);
            const float cn = fg * cstate[c] + ig * jg;
            cstate[c] = cn;
            const float hn = og * tanhf_fast(cn);
            const int unit = (member * UG + slice) * 32 + fu;
            __hip_atomic_store(a.hout + (row0 + r) * (2 * HP) + dir * HP + unit, hn, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            if (SAVE) {
                float* rv = a.resv + (row0 + r) * (2 * 5 * HP) + dir * 5 * HP + unit;
                rv[0 * HP] = ig, rv[1 * HP] = jg, rv[2 * HP] = fg, rv[3 * HP] = og, rv[4 * HP] = cn;
            }
        }

        // ---- publish: every wave waits for its own (write-through) stores, then one increment per workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // also: nobody overwrites `part` before all cells of this step are read
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (member == 0 && tid == 0 && !dead) reset_counter_when_done(ctr, (unsigned)S * (unsigned)T, a.spin_ticks, a.sync);
}

// Finer split for the smallest batches (S = 16 or 32 workgroups per (tile, direction)): at S = 8 the 128
// MFMAs per SIMD and step are 3.2 us of an 7.7 us step, and a batch of 32 .. 256 utterances leaves most of the
// chip idle, so the slice of a workgroup shrinks to UW = 256 / S = 16 or 8 hidden units.  Its 4 x UW gate
// columns form NT = 2 or 1 MFMA column tiles (gates (i, j) and (f, o) of 16 units, or all four gates of 8
// units -- the B fragments are gathered accordingly from the same packed Wh, once); the eight waves split
// the reduction as before (4 k-groups each), so a wave issues 16 NT MFMAs per step instead of 64.
// XCH: the members exchange h through a buffer of their own, laid out [step][group][member][32 utterances][UW units]:
// a member's contribution to a step is ONE contiguous block (1 or 2 KB), so its write-through stores fill whole
// 128-byte lines (in hout a member owns 32- or 64-byte pieces of lines: to a line no cache holds, such a store is
// acknowledged only after the L2 has fetched the rest, 0.78 -> 0.86 ms per layer at 32 utterances on the cold buffers a
// real step has), a reading wave's fragment load is one contiguous block, and the invariant of cacheable loads holds
// (coherent_load4_issue): nothing touches a line before it is complete.  hout is written beside it with plain stores
// nobody waits for.  Without XCH the exchange runs through hout itself: device-scope loads, lines touched ahead.
// STAMPS (diagnostic instantiation, selected while avsi_diag_cs_stamps holds a buffer; tools/rec_fine_stamps.py): waves 0 and 1
// of the first 32 workgroups record the 100 MHz wall clock at eight points of steps 64 .. 71 -- top of the step, counter seen,
// barrier passed, h fragments landed, MFMAs + park + barrier, cell done and stores issued, h store acknowledged, published.
template <int NT, bool SAVE, bool XCH, bool STAMPS = false>
__global__ __launch_bounds__(512, 2) void blstm_rec_fwd_coop_fine_kernel(const CoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = reinterpret_cast<float*>(smem);      // [wave][tile][column][row]
    __shared__ int dead;
    constexpr int S = 32 / NT;                         // members per group
    constexpr int UW = 256 / S;                        // hidden units of this workgroup
    constexpr int NG = 32 / UW;                        // gates per column tile
    constexpr int PER = S / 8;                         // members per 32-unit slice
    constexpr int QPW = 4;

    const int xcd = blockIdx.x % AVSI_NUM_XCD, kk = blockIdx.x / AVSI_NUM_XCD;
    const int member = kk % S;
    const int group = (kk / S) * AVSI_NUM_XCD + xcd;
    if (group >= a.ngroups) return;
    // behind a launch that gave up a bounded wait every result is void (sticky status word): leave at once instead of
    // waiting out the time bound on step counters that launch may have left behind
    if (avsi_launch_is_void(a.sync)) return;
    const int dir = group & 1;
    const int b0 = (a.tile0 + (group >> 1)) * 32;
    const int T = a.T, Bp = a.Bp;
    const int w = member / PER, u0 = (member % PER) * UW;   // slice and first unit inside it

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ks = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    if (tid == 0) dead = 0;

    // this wave's piece of Wh: column li of tile tl = gate tl * NG + li / UW, unit u0 + li % UW
    float4 wreg[QPW][NT];
    {
        const float4* wp = reinterpret_cast<const float4*>(a.whp) + (size_t)(dir * 8 + w) * (32 * 4 * 64);
#pragma unroll
        for (int q = 0; q < QPW; ++q)
#pragma unroll
            for (int tl = 0; tl < NT; ++tl)
                wreg[q][tl] = wp[((ks * QPW + q) * 4 + tl * NG + li / UW) * 64 + hi * 32 + u0 + li % UW];
    }

    // finishing role: one cell (row frow, unit fu) per lane; with UW = 8 the upper half of the block idles
    const int fu = tid % UW, frow = tid / UW;
    const bool fin = frow < 32;
    float cstate = 0.f;

    unsigned* ctr = a.sync + CTR_STRIDE * (1 + 2 * a.tile0 + group);
    // exchange layout: [step][global group][256 units as S member blocks][32 rows][UW]
    const size_t xgroups = (size_t)2 * a.xtiles;
    float* xbase = XCH ? a.xch + (size_t)(2 * (a.tile0 - a.xtile0) + group) * (32 * HP) : nullptr;
    __syncthreads();

    auto stamp = [&](int step, int phase) {
        if (STAMPS && (tid & 63) == 0 && tid < 128 && blockIdx.x < 32 && step >= 64 && step < 72)
            a.stamps[((blockIdx.x * 8 + (step - 64)) * 2 + (tid >> 6)) * 8 + phase] = wall_clock64();
    };
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        const int tprev = dir ? t + 1 : t - 1;
        const size_t row0 = (size_t)t * Bp + b0;
        stamp(step, 0);

        float xz[4];
        float touched = 0.f;
        if (fin) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                xz[g] = a.xproj[(row0 + frow) * (2 * GP) + dir * GP + w * 128 + g * 32 + u0 + fu];
            if (!XCH) {   // Touch the line this lane will store to two steps from now.  A workgroup writes 32- or 64-byte pieces
                // of a line; to a line no cache holds, the write-through store was acknowledged later (the L2 fetches
                // the rest of the line first).  The loaded value is never used and the readers of this kernel use
                // device-scope loads only, so the early (incomplete) copy this leaves in the L1 and L2 is never read;
                // an `sc1` touch (no L1 copy) did not help at all (0.92 ms against 0.78), a scalar-cache touch less,
                // and `buffer_inv sc1` before cacheable reads cost 4 ms.  The register stays reserved until the
                // end-of-step wait: the compiler does not know the load is still in flight.
                const int st2 = step + 2 < T ? step + 2 : T - 1;
                const int t2 = dir ? (T - 1 - st2) : st2;
                asm volatile("global_load_dword %0, %1, off" : "=v"(touched) : "v"(a.hout + ((size_t)t2 * Bp + b0 + frow) * (2 * HP) + dir * HP + w * 32 + u0 + fu) : "memory");
            }
        }

        f32x16 acc[NT];
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tl][r] = 0.f;

        if (step > 0) {
            if (tid == 0 && !dead) {
                const unsigned want = (unsigned)S * (unsigned)step;
                unsigned polls = 0;
                long long t0 = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (avsi_spin_expired(polls, t0, a.spin_ticks, a.sync)) {
                        dead = 1;
                        atomicExch(a.sync, 1u);
                        break;
                    }
                }
            }
            stamp(step, 1);
            __syncthreads();
            stamp(step, 2);
            v4f af[4];
            if (XCH) {
                // k group q of this wave = units 32 ks + 8 q ..: block of member (32 ks + 8 q) / UW, offset (8 q) % UW
                const float* xp = xbase + (size_t)(step - 1) * xgroups * (32 * HP);
                if (a.coherent) {       // triage switch: the same loads served at the memory side
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int unit0 = 32 * ks + 8 * q;
                        coherent_load4_issue<false>(af[q], xp + ((unit0 / UW) * 32 + li) * UW + unit0 % UW + 4 * hi, 0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int unit0 = 32 * ks + 8 * q;
                        coherent_load4_issue<true>(af[q], xp + ((unit0 / UW) * 32 + li) * UW + unit0 % UW + 4 * hi, 0);
                    }
                }
            } else {
                const float* hp = a.hout + ((size_t)tprev * Bp + b0 + li) * (2 * HP) + dir * HP + 4 * hi;
#pragma unroll
                for (int q = 0; q < 4; ++q) agent_load4_issue(af[q], hp + 8 * ks * QPW, 32 * q);
            }
            coherent_wait(af);
            if (XCH && a.noack) {
                auto poisoned = [&]() {
                    bool bad = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        bad |= __float_as_uint(af[q].x) == COOP_POISON || __float_as_uint(af[q].y) == COOP_POISON ||
                               __float_as_uint(af[q].z) == COOP_POISON || __float_as_uint(af[q].w) == COOP_POISON;
                    return __builtin_amdgcn_ballot_w64(bad) != 0;       // one verdict per wave
                };
                unsigned polls = 0;
                long long t0 = 0;
                while (poisoned()) {
                    const float* xp = xbase + (size_t)(step - 1) * xgroups * (32 * HP);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int unit0 = 32 * ks + 8 * q;
                        coherent_load4_issue<false>(af[q], xp + ((unit0 / UW) * 32 + li) * UW + unit0 % UW + 4 * hi, 0);
                    }
                    coherent_wait(af);
                    if (avsi_spin_expired(polls, t0, a.spin_ticks, a.sync)) {
                        atomicExch(a.sync, 1u);       // results are void from here on; the step counters still move, nobody hangs
                        break;
                    }
                }
            }
            stamp(step, 3);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tl = 0; tl < NT; ++tl) {
                        const float4 b = wreg[q][tl];
                        const float av = s == 0 ? af[q].x : s == 1 ? af[q].y : s == 2 ? af[q].z : af[q].w;
                        const float bv = s == 0 ? b.x : s == 1 ? b.y : s == 2 ? b.z : b.w;
                        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tl], 0, 0, 0);
                    }
        }

        // park the partial tiles: part[ks][tile][column li][row], rows (r&3) + 8 (r>>2) + 4 hi
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(part + ((ks * NT + tl) * 32 + li) * PSTRIDE + 8 * j + 4 * hi) =
                    make_float4(acc[tl][4 * j], acc[tl][4 * j + 1], acc[tl][4 * j + 2], acc[tl][4 * j + 3]);
        __syncthreads();
        stamp(step, 4);

        if (fin) {
            float z[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float sum = xz[g];
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += part[((k * NT + g / NG) * 32 + (g % NG) * UW + fu) * PSTRIDE + frow];
                z[g] = sum;
            }
            const float ig = sigmoidf_fast(z[0]), jg = tanhf_fast(z[1]), fg = sigmoidf_fast(z[2]), og = sigmoidf_fast(z[3]);
            const float cn = fg * cstate + ig * jg;
            cstate = cn;
            const float hn = og * tanhf_fast(cn);
            const int unit = w * 32 + u0 + fu;
            if (XCH) {      // the exchange copy first (the only store the publication waits for), hout beside it
                __hip_atomic_store(xbase + (size_t)step * xgroups * (32 * HP) + (member * 32 + frow) * UW + fu, hn,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // compiler barrier: the counted wait below assumes the exchange store is ISSUED before the plain stores
                // that follow; nothing else stops the compiler from moving a non-atomic store across a relaxed atomic one
                asm volatile("" ::: "memory");
                a.hout[(row0 + frow) * (2 * HP) + dir * HP + unit] = hn;
            } else {
                __hip_atomic_store(a.hout + (row0 + frow) * (2 * HP) + dir * HP + unit, hn, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" ::: "memory");
            }
            if (SAVE) {
                float* rv = a.resv + (row0 + frow) * (2 * 5 * HP) + dir * 5 * HP + unit;
                rv[0 * HP] = ig, rv[1 * HP] = jg, rv[2 * HP] = fg, rv[3 * HP] = og, rv[4 * HP] = cn;
            }
        }
        // Only the h store has to be complete before the counter moves.  Stores complete in issue order, so with the
        // five reserve stores issued BEHIND it the wait leaves those in flight.  (Measured: no difference, neither on
        // warm buffers nor inside a training step -- kept because it is the weaker, sufficient condition.)
        stamp(step, 5);
        // (STAMPS: the stamps' own stores sit in the same queue -- the diagnostic build waits for everything)
        if (XCH && a.noack) {
            // no wait: the stores are issued (the barrier below orders the ISSUE of every lane's store before the counter),
            // their acknowledgement is collected by the next step's wait for its fragment loads
        } else if (STAMPS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (SAVE && XCH) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (SAVE) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (XCH) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(touched));
        stamp(step, 6);
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(step, 7);
    }
    if (member == 0 && tid == 0 && !dead) reset_counter_when_done(ctr, (unsigned)S * (unsigned)T, a.spin_ticks, a.sync);
}

// Half-row form of the 32-way kernel (round 5, split code 64; up to 64 utterances, where CUs are idle anyway): the 32 utterances
// of a tile are two independent groups of 16 rows with 32 members each.  The stamps of the 32-row kernel
// (tools/rec_fine_stamps.py) put the MFMA phase at 0.85 - 1.0 us of a 2.7 us step -- a CU issues 128 v_mfma_f32_32x32x2 per step,
// two waves to a SIMD -- and a member cannot be made narrower than 8 units without doubling the fan-in of the exchange.  With 16
// rows the product is 16 x 32 per member: v_mfma_f32_16x16x4_f32, two column tiles (gates (i, j) and (f, o) of the 8 units),
// 16 short MFMAs per wave and step instead of 16 long ones.  Lane l = (row or column l % 16, k slot l / 16); a lane's eight k
// values are eight CONSECUTIVE units (two 16-byte fragment loads from the exchange copy; any bijection of k works as long as
// A and B agree), its weights 2 x 8 registers gathered once from the packed Wh.  Exchange layout [step][group][member][16 rows]
// [8 units]; always through the exchange copy (no hout-exchange form).  128 finishing lanes: one cell each.
template <bool SAVE>
__global__ __launch_bounds__(512, 2) void blstm_rec_fwd_coop_half_kernel(const CoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = reinterpret_cast<float*>(smem);      // [wave][tile][column 16][row 16 (+4)]
    __shared__ int dead;
    constexpr int S = 32, UW = 8, PER = 4, RP = 20;
    typedef float f32x4v __attribute__((ext_vector_type(4)));

    const int xcd = blockIdx.x % AVSI_NUM_XCD, kk_ = blockIdx.x / AVSI_NUM_XCD;
    const int member = kk_ % S;
    const int group = (kk_ / S) * AVSI_NUM_XCD + xcd;          // (tile, half, direction) of this launch
    if (group >= a.ngroups) return;
    if (avsi_launch_is_void(a.sync)) return;
    const int dir = group & 1, half = (group >> 1) & 1;
    const int b0 = (a.tile0 + (group >> 2)) * 32 + 16 * half;
    const int T = a.T, Bp = a.Bp;
    const int w = member / PER, u0 = (member % PER) * UW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ks = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, kq = lane >> 4;
    if (tid == 0) dead = 0;

    // this lane's weights: Wh[k = 32 ks + 8 kq + kk][gate 2 tl + c16 / 8][unit 32 w + u0 + c16 % 8], kk = 0 .. 7
    // whp [2][8 w][32 q][4 g][64 lane][4 s] = Wh[k = 8 q + 4 (lane / 32) + s][gate g][unit 32 w + lane % 32]
    float4 wreg[2][2];                                  // [tile][k half]
    {
        const float4* wp = reinterpret_cast<const float4*>(a.whp) + (size_t)(dir * 8 + w) * (32 * 4 * 64);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
                wreg[tl][kh] = wp[((4 * ks + kq) * 4 + 2 * tl + c16 / UW) * 64 + kh * 32 + u0 + c16 % UW];
    }

    // finishing role: one cell (row frow, unit fu) per lane of the first two waves
    const int fu = tid % UW, frow = tid / UW;
    const bool fin = frow < 16;
    float cstate = 0.f;

    unsigned* ctr = a.sync + CTR_STRIDE * (1 + 4 * a.tile0 + group);
    // exchange layout: [step][global group = (tile, half, direction)][32 members][16 rows][8 units]
    const size_t xgroups = (size_t)4 * a.xtiles;
    float* xbase = a.xch + (size_t)(4 * (a.tile0 - a.xtile0) + group) * (16 * HP);
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        const size_t row0 = (size_t)t * Bp + b0;

        float xz[4];
        if (fin) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                xz[g] = a.xproj[(row0 + frow) * (2 * GP) + dir * GP + w * 128 + g * 32 + u0 + fu];
        }

        f32x4v acc[2];
        acc[0] = acc[1] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            if (tid == 0 && !dead) {
                const unsigned want = (unsigned)S * (unsigned)step;
                unsigned polls = 0;
                long long t0 = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (avsi_spin_expired(polls, t0, a.spin_ticks, a.sync)) {
                        dead = 1;
                        atomicExch(a.sync, 1u);
                        break;
                    }
                }
            }
            __syncthreads();
            // units 32 ks + 8 kq .. + 7 of row c16 = block of member 4 ks + kq, row c16: 32 contiguous bytes
            const float* xp = xbase + (size_t)(step - 1) * xgroups * (16 * HP) + ((4 * ks + kq) * 16 + c16) * UW;
            v4f af[2];
            if (a.coherent) {
                coherent_load4_issue<false>(af[0], xp, 0);
                coherent_load4_issue<false>(af[1], xp, 16);
            } else {
                coherent_load4_issue<true>(af[0], xp, 0);
                coherent_load4_issue<true>(af[1], xp, 16);
            }
            coherent_wait(af);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float av = s4 == 0 ? af[kh].x : s4 == 1 ? af[kh].y : s4 == 2 ? af[kh].z : af[kh].w;
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        const float4 b = wreg[tl][kh];
                        const float bv = s4 == 0 ? b.x : s4 == 1 ? b.y : s4 == 2 ? b.z : b.w;
                        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[tl], 0, 0, 0);
                    }
                }
        }
        // park: D[row 4 kq + i][column c16] -> part[(ks * 2 + tl) * 16 + c16][row]
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
            *reinterpret_cast<float4*>(part + ((ks * 2 + tl) * 16 + c16) * RP + 4 * kq) =
                make_float4(acc[tl][0], acc[tl][1], acc[tl][2], acc[tl][3]);
        __syncthreads();

        if (fin) {
            float z[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float sum = xz[g];
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += part[((k * 2 + g / 2) * 16 + (g % 2) * UW + fu) * RP + frow];
                z[g] = sum;
            }
            const float ig = sigmoidf_fast(z[0]), jg = tanhf_fast(z[1]), fg = sigmoidf_fast(z[2]), og = sigmoidf_fast(z[3]);
            const float cn = fg * cstate + ig * jg;
            cstate = cn;
            const float hn = og * tanhf_fast(cn);
            const int unit = w * 32 + u0 + fu;
            __hip_atomic_store(xbase + (size_t)step * xgroups * (16 * HP) + (member * 16 + frow) * UW + fu, hn, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::: "memory");      // the exchange store is ISSUED before the plain stores (counted wait below)
            a.hout[(row0 + frow) * (2 * HP) + dir * HP + unit] = hn;
            if (SAVE) {
                float* rv = a.resv + (row0 + frow) * (2 * 5 * HP) + dir * 5 * HP + unit;
                rv[0 * HP] = ig, rv[1 * HP] = jg, rv[2 * HP] = fg, rv[3 * HP] = og, rv[4 * HP] = cn;
            }
        }
        if (SAVE) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (member == 0 && tid == 0 && !dead) reset_counter_when_done(ctr, (unsigned)S * (unsigned)T, a.spin_ticks, a.sync);
}

// LDS the fine kernels REQUEST (they use 16 - 37 KB): more than half of a CU's 160 KB so that one member sits on a CU.
// AVSI_COOP_FINE_LDS_KB (96 .. 160) asks for more -- round-5 experiment: at 120 KB no workgroup of the weight-gradient GEMMs
// (49 KB) that run beside the BPTT kernels on side streams fits on a member's CU, and the BPTT kernels of a training step at
// 32 utterances do get faster (2.60 -> 2.36 ms for the three layers: a co-resident GEMM workgroup shares the member's SIMDs and
// matrix pipes), but the GEMMs, confined to the other 192 CUs, finish so much later that the step is slower (5.68 -> 5.94 ms).
// So 96 stays.
static size_t fine_lds_bytes() {
    static const size_t v = [] {
        const char* e = getenv("AVSI_COOP_FINE_LDS_KB");
        long kb = e ? atol(e) : 96;
        kb = kb < 96 ? 96 : (kb > 160 ? 160 : kb);
        return (size_t)kb * 1024;
    }();
    return v;
}

template <bool SAVE>
int launch_coop_half(const CoopArgs& a, hipStream_t st) {
    const size_t lds = fine_lds_bytes();
    (void)hipFuncSetAttribute((const void*)blstm_rec_fwd_coop_half_kernel<SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = (int)avsi_ceil_div(a.ngroups, AVSI_NUM_XCD) * AVSI_NUM_XCD * 32;
    hipLaunchKernelGGL((blstm_rec_fwd_coop_half_kernel<SAVE>), dim3(blocks), dim3(512), lds, st, a);
    return avsi_launch_status();
}

template <int NT, bool SAVE, bool XCH>
int launch_coop_fine_x(const CoopArgs& a, hipStream_t st) {
    constexpr int S = 32 / NT;
    // > half of the CU's LDS on purpose: one workgroup per CU, so the members of a group spread over S CUs
    const size_t lds = fine_lds_bytes();
    static_assert((size_t)8 * NT * 32 * PSTRIDE * sizeof(float) <= 96 * 1024, "partial tiles must fit");
    (void)hipFuncSetAttribute((const void*)blstm_rec_fwd_coop_fine_kernel<NT, SAVE, XCH>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = (int)avsi_ceil_div(a.ngroups, AVSI_NUM_XCD) * AVSI_NUM_XCD * S;
    if (a.stamps && NT == 1 && !SAVE && XCH) {        // diagnostic instantiation (avsi_diag_cs_stamps): the 32-way inference kernel
        (void)hipFuncSetAttribute((const void*)blstm_rec_fwd_coop_fine_kernel<1, false, true, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((blstm_rec_fwd_coop_fine_kernel<1, false, true, true>), dim3(blocks), dim3(512), lds, st, a);
        return avsi_launch_status();
    }
    hipLaunchKernelGGL((blstm_rec_fwd_coop_fine_kernel<NT, SAVE, XCH>), dim3(blocks), dim3(512), lds, st, a);
    return avsi_launch_status();
}
template <int NT, bool SAVE>
int launch_coop_fine(const CoopArgs& a, hipStream_t st) {
    return a.xch ? launch_coop_fine_x<NT, SAVE, true>(a, st) : launch_coop_fine_x<NT, SAVE, false>(a, st);
}

template <int S, bool SAVE>
int launch_coop(const CoopArgs& a, hipStream_t st) {
    const size_t lds = (size_t)PART_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute((const void*)blstm_rec_fwd_coop_kernel<S, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    const int blocks = (int)avsi_ceil_div(a.ngroups, AVSI_NUM_XCD) * AVSI_NUM_XCD * S;
    hipLaunchKernelGGL((blstm_rec_fwd_coop_kernel<S, SAVE>), dim3(blocks), dim3(512), lds, st, a);
    return avsi_launch_status();
}

}  // namespace

unsigned long long* avsi_cs_stamps_buffer();     // blstm_fwd_cs.hip: the diagnostic buffer of avsi_diag_cs_stamps (null: no stamps)

// word 0: status, then step counters: one per (tile, direction), two for the half-tile BPTT kernel
extern "C" size_t avsi_blstm_rec_fwd_coop_workspace_bytes(int Bp) {
    return (size_t)CTR_STRIDE * (1 + 4 * (Bp > 0 ? Bp / 32 : 0)) * sizeof(unsigned);
}

// h of a whole launch in the exchange layout of the fine forward kernels: as many bytes as hout
extern "C" size_t avsi_blstm_rec_fwd_coop_exchange_bytes(int T, int Bp) {
    return (T > 0 && Bp > 0) ? (size_t)T * (size_t)Bp * (2 * HP) * sizeof(float) : 0;
}
// ... and dz of a whole launch for the fine BPTT kernels
extern "C" size_t avsi_blstm_rec_bwd_coop_exchange_bytes(int T, int Bp) {
    return (T > 0 && Bp > 0) ? (size_t)T * (size_t)Bp * (2 * 4 * HP) * sizeof(float) : 0;
}

// AVSI_COOP_COHERENT=1 (read at every call): the one switch that takes EVERY cooperative exchange back to device-scope
// loads -- the kernels that read exchanged bytes with cacheable loads under an invariant (4- / 8-way BPTT: dz lines
// written once before their counter moves and touched by nobody earlier; fine kernels: the exchange layout) then use
// `sc0 sc1` loads like the rest.  For triage: a result that changes with this switch is a stale-line bug.
static int coop_coherent() {
    const char* e = getenv("AVSI_COOP_COHERENT");
    return e && e[0] == '1';
}

// Tiles per launch: the whole launch must be resident (one workgroup per CU) on the `max_cus` compute units the
// caller grants it (<= 0: the chip).  A process that keeps other kernels in flight beside the recurrence -- RCCL
// collectives under data parallelism, independent batches on other streams -- leaves those CUs out: peers of a
// group that cannot become resident would otherwise wait for each other until the bounded spin gives up.
// Whole (tile, direction) groups, in multiples of the XCD count when there are that many (members of a group sit
// 8 block ids apart).  0 = not even one tile fits.
static int coop_tiles_per_launch(int split, int max_cus) {
    const int cus = (max_cus <= 0 || max_cus > AVSI_NUM_CU) ? AVSI_NUM_CU : max_cus;
    int groups = cus / split;
    if (groups >= AVSI_NUM_XCD) groups = groups / AVSI_NUM_XCD * AVSI_NUM_XCD;
    return groups / 2;
}

extern "C" int avsi_blstm_rec_fwd_coop_rows_f32(const float* xproj, const float* whp, float* hout, float* reserve, int T, int Bp,
                                                int split, int first_row, int rows, int max_cus, void* workspace,
                                                size_t workspace_bytes, void* stream) {
    if (!xproj || !whp || !hout || T <= 0 || Bp <= 0 || (Bp & 31)) return AVSI_ERR_INVALID_ARG;
    if (split != 4 && split != 8 && split != 16 && split != 32 && split != 64) return AVSI_ERR_INVALID_ARG;
    if (first_row < 0 || rows <= 0 || (first_row & 31) || (rows & 31) || first_row + rows > Bp) return AVSI_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < avsi_blstm_rec_fwd_coop_workspace_bytes(Bp)) return AVSI_ERR_WORKSPACE;
    if (coop_tiles_per_launch(split, max_cus) < 1) return AVSI_ERR_UNSUPPORTED;   // 2 * split workgroups do not fit max_cus
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    // no memset: the caller zeroes the workspace once, every launch leaves its counters at zero again
    // (reset_counter_when_done) and never touches the sticky status word
    // every member of a group must be resident while its peers wait for it: batches beyond one chip-full of
    // groups run as consecutive launches over tile ranges
    const int tbeg = first_row / 32, tiles = tbeg + rows / 32, per = coop_tiles_per_launch(split, max_cus);
    // a workspace that also holds avsi_blstm_rec_fwd_coop_exchange_bytes(T, Bp) at AVSI_COOP_EXCHANGE_OFFSET switches the fine
    // splits (16, 32) to the exchange layout (see blstm_rec_fwd_coop_fine_kernel)
    // (at a FIXED offset: the counters of a later, larger batch must not land on old exchange data)
    float* xch = (split >= 16 && workspace_bytes >= AVSI_COOP_EXCHANGE_OFFSET + avsi_blstm_rec_fwd_coop_exchange_bytes(T, rows))
                     ? reinterpret_cast<float*>(static_cast<char*>(workspace) + AVSI_COOP_EXCHANGE_OFFSET) : nullptr;
    // the exchange copy sits at a fixed offset: the counters of this batch must end in front of it
    if (xch && avsi_blstm_rec_fwd_coop_workspace_bytes(Bp) > AVSI_COOP_EXCHANGE_OFFSET) return AVSI_ERR_WORKSPACE;
    if (split == 64 && !xch) return AVSI_ERR_WORKSPACE;        // the half-row kernel has no form that exchanges through hout
    // AVSI_COOP_NOACK=1 (read at every call): the exchange copy of this call preset to the poison pattern, publication without
    // the store acknowledgement (see COOP_POISON)
    const char* na = getenv("AVSI_COOP_NOACK");
    const int noack = (xch && na && na[0] == '1' && !coop_coherent()) ? 1 : 0;
    if (noack && hipMemsetAsync(xch, 0xFF, avsi_blstm_rec_fwd_coop_exchange_bytes(T, rows), st) != hipSuccess) return AVSI_ERR_LAUNCH;
    for (int tile0 = tbeg; tile0 < tiles; tile0 += per) {
        const int nt = tiles - tile0 < per ? tiles - tile0 : per;
        CoopArgs a{xproj, whp, hout, reserve, (unsigned*)workspace, T, Bp, 2 * nt, tile0, xch, tbeg, rows / 32, coop_coherent(), avsi_coop_spin_ticks(), avsi_cs_stamps_buffer(), noack};
        int rc;
        if (split == 64) {
            a.ngroups = 4 * nt;                               // (tile, half, direction)
            a.noack = 0;
            rc = reserve ? launch_coop_half<true>(a, st) : launch_coop_half<false>(a, st);
        } else if (split == 32)
            rc = reserve ? launch_coop_fine<1, true>(a, st) : launch_coop_fine<1, false>(a, st);
        else if (split == 16)
            rc = reserve ? launch_coop_fine<2, true>(a, st) : launch_coop_fine<2, false>(a, st);
        else if (split == 8)
            rc = reserve ? launch_coop<8, true>(a, st) : launch_coop<8, false>(a, st);
        else
            rc = reserve ? launch_coop<4, true>(a, st) : launch_coop<4, false>(a, st);
        if (rc != AVSI_OK) return rc;
    }
    return AVSI_OK;
}

extern "C" int avsi_blstm_rec_fwd_coop_f32(const float* xproj, const float* whp, float* hout, float* reserve, int T, int Bp,
                                           int split, int max_cus, void* workspace, size_t workspace_bytes, void* stream) {
    return avsi_blstm_rec_fwd_coop_rows_f32(xproj, whp, hout, reserve, T, Bp, split, 0, Bp, max_cus, workspace, workspace_bytes,
                                            stream);
}

// ==========================================================================================
// Cooperative BPTT (small batches): the gradient of the kernel above, same decomposition.
//   dh = dH_out[t] + dz_next . Wh^T   reduces over the 1024 packed gate columns of the step
// processed just before.  Workgroup m of a group owns 8 / S of the 32-unit slices of dh; inside
// it the 1024-long reduction is split over S waves per slice (128 / S k-groups of 8 each), whose
// piece of Wh^T (whbT fragment order of blstm_bwd.hip) stays in registers for all T steps.  The
// group exchanges dz through the dz output buffer itself (device-coherent stores / loads + the
// same per-step counter protocol); every lane then finishes two (four) cells: sum of the S
// partials, gate gradients from the forward reserve, dc_next in registers.
// ==========================================================================================
namespace {

struct CoopBwdArgs {
    const float* dhout;
    const float* resv;
    const float* whbT;
    float* dz;
    unsigned* sync;
    int T, Bp, ngroups;
    int tile0;
    float* xch;        // fine kernel: dz in exchange layout [step][group x half][member][gate][rows][16 units], or null
    int coherent;      // AVSI_COOP_COHERENT=1 (see CoopArgs)
    long long spin_ticks;
    unsigned long long* stamps;   // diagnostics (STAMPS instantiation of the half-tile fine kernel, avsi_diag_cs_stamps)
};

constexpr int BPART_FLOATS = 8 * 32 * PSTRIDE;      // [wave][unit][row]

template <int S>
__global__ __launch_bounds__(512, 2) void blstm_rec_bwd_coop_kernel(const CoopBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float part[BPART_FLOATS];
    __shared__ int dead;
    constexpr int UG = 8 / S;
    constexpr int QPW = 128 / S;                       // k groups of 8 (packed gate columns) per wave
    constexpr int CPL = 2 * UG;

    const int xcd = blockIdx.x % AVSI_NUM_XCD, kk = blockIdx.x / AVSI_NUM_XCD;
    const int member = kk % S;
    const int group = (kk / S) * AVSI_NUM_XCD + xcd;
    if (group >= a.ngroups) return;
    // behind a launch that gave up a bounded wait every result is void (sticky status word): leave at once instead of
    // waiting out the time bound on step counters that launch may have left behind
    if (avsi_launch_is_void(a.sync)) return;
    const int dir = group & 1;
    const int b0 = (a.tile0 + (group >> 1)) * 32;
    const int T = a.T, Bp = a.Bp;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int ugl = v / S, ks = v % S;
    const int wq = member * UG + ugl;
    if (tid == 0) dead = 0;

    // this wave's piece of Wh^T: whbT [2][8 w][128 q][64 lane][4 s]
    float4 wreg[QPW];
    {
        const float4* wb = reinterpret_cast<const float4*>(a.whbT) + (size_t)(dir * 8 + wq) * (128 * 64) + lane;
#pragma unroll
        for (int q = 0; q < QPW; ++q) wreg[q] = wb[(ks * QPW + q) * 64];
    }

    const int fu = tid & 31, frow = (tid >> 5) * 2;
    float dcn[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) dcn[c] = 0.f;
    unsigned* ctr = a.sync + CTR_STRIDE * (1 + 2 * a.tile0 + group);
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : (T - 1 - s);             // fw walks T-1 .. 0, bw walks 0 .. T-1
        const int tnext = dir ? t - 1 : t + 1;           // the step processed just before this one
        const int tp = dir ? t + 1 : t - 1;              // forward-previous step (owner of c_prev)
        const bool has_prev = dir ? (t + 1 < T) : (t > 0);
        const size_t row0 = (size_t)t * Bp + b0;

        // inputs of this lane's cells (in flight during the wait and the MFMAs)
        float dh[CPL], gi[CPL], gj[CPL], gf[CPL], go[CPL], cc[CPL], cp[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int unit = (member * UG + (c >> 1)) * 32 + fu;
            const size_t row = row0 + frow + (c & 1);
            const float* rv = a.resv + row * (2 * 5 * HP) + dir * 5 * HP + unit;
            dh[c] = a.dhout[row * (2 * HP) + dir * HP + unit];
            gi[c] = rv[0 * HP], gj[c] = rv[1 * HP], gf[c] = rv[2 * HP], go[c] = rv[3 * HP], cc[c] = rv[4 * HP];
            cp[c] = has_prev ? a.resv[((size_t)tp * Bp + b0 + frow + (c & 1)) * (2 * 5 * HP) + dir * 5 * HP + 4 * HP + unit] : 0.f;
        }

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (s > 0) {
            if (tid == 0 && !dead) {
                const unsigned want = (unsigned)S * (unsigned)s;
                unsigned polls = 0;
                long long t0 = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (avsi_spin_expired(polls, t0, a.spin_ticks, a.sync)) {
                        dead = 1;
                        atomicExch(a.sync, 1u);
                        break;
                    }
                }
            }
            __syncthreads();
            const float* zp = a.dz + ((size_t)tnext * Bp + b0 + li) * (2 * GP) + dir * GP + 4 * hi;
            // fragments in batches of 8 (the weights already hold 64 / 128 registers)
#pragma unroll
            for (int c0 = 0; c0 < QPW; c0 += 8) {
                v4f af[8];
                if (a.coherent) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) coherent_load4_issue<false>(af[q], zp + 8 * ks * QPW, 32 * (c0 + q));
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) coherent_load4_issue<true>(af[q], zp + 8 * ks * QPW, 32 * (c0 + q));
                }
                coherent_wait(af);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 b = wreg[c0 + q];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, b.w, acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(part + (v * 32 + li) * PSTRIDE + 8 * j + 4 * hi) =
                make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
        __syncthreads();

#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int slice = c >> 1, r = frow + (c & 1);
            float d = dh[c];
#pragma unroll
            for (int k = 0; k < S; ++k) d += part[((slice * S + k) * 32 + fu) * PSTRIDE + r];
            const float ig = gi[c], jg = gj[c], fg = gf[c], og = go[c];
            const float tc = 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * cc[c])) - 1.f;
            const float dc = d * og * (1.f - tc * tc) + dcn[c];
            dcn[c] = dc * fg;
            float* zo = a.dz + (row0 + r) * (2 * GP) + dir * GP + (member * UG + slice) * 128 + fu;
            __hip_atomic_store(zo + 0, dc * jg * ig * (1.f - ig), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(zo + 32, dc * ig * (1.f - jg * jg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(zo + 64, dc * cp[c] * fg * (1.f - fg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(zo + 96, d * tc * og * (1.f - og), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (member == 0 && tid == 0 && !dead) reset_counter_when_done(ctr, (unsigned)S * (unsigned)T, a.spin_ticks, a.sync);
}

// Finer split of the BPTT kernel (S = 16: a workgroup owns 16 units of dh), the counterpart of
// blstm_rec_fwd_coop_fine_kernel.  A 32x32x2 tile would leave half its columns empty, so this kernel uses
// v_mfma_f32_16x16x4_f32 (same rate: 64 flop / cycle / SIMD): two 16-row tiles x 16 units, the eight waves
// split the 1024 packed gate columns (128 each); a lane's float4 of dz covers four consecutive columns, used
// as the k-slices of four MFMAs, and its 32 weights Wh[unit][those columns] are gathered once from the same
// whbT operand.  64 short MFMAs per wave and step instead of 64 long ones.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// RH = 2: the 32 utterances of a tile are two independent halves of 16 with 16 workgroups each (32 per
// (tile, direction)): one MFMA row tile and half the dz rows per workgroup.
// XCH: as in the forward kernel, the members exchange dz through a copy of their own, [step][group x half][member]
// [gate][rows][16 units]: a member's four gate blocks of a step are written as whole lines (in dz it owns 64-byte pieces),
// a reading wave's fragment load is one contiguous KB, cacheable loads are valid by construction; dz itself is written
// beside it with plain stores nobody waits for.  Without XCH: exchange through dz, lines touched ahead, device-scope loads.
// STAMPS: as in the forward kernel (tools/rec_fine_stamps.py bwd): top, counter seen, barrier, dz fragments landed, MFMAs + park +
// barrier, cell done and stores issued, exchange stores acknowledged, published.
template <int RH, bool XCH, bool STAMPS = false>
__global__ __launch_bounds__(512, 2) void blstm_rec_bwd_coop_fine_kernel(const CoopBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    float* part = reinterpret_cast<float*>(smem_b);    // [wave][unit 16][row 32 (+4 pad)]
    __shared__ int dead;
    constexpr int S = 16, UW = 16;                      // members that exchange with each other, units per member
    constexpr int NR = 2 / RH;                         // 16-row MFMA tiles per workgroup

    const int xcd = blockIdx.x % AVSI_NUM_XCD, kk = blockIdx.x / AVSI_NUM_XCD;
    const int mem_all = kk % (S * RH);
    const int member = mem_all % S, half = mem_all / S;
    const int group = (kk / (S * RH)) * AVSI_NUM_XCD + xcd;
    if (group >= a.ngroups) return;
    // behind a launch that gave up a bounded wait every result is void (sticky status word): leave at once instead of
    // waiting out the time bound on step counters that launch may have left behind
    if (avsi_launch_is_void(a.sync)) return;
    const int dir = group & 1;
    const int b0 = (a.tile0 + (group >> 1)) * 32 + half * 16;
    const int T = a.T, Bp = a.Bp;
    const int w = member >> 1, u0 = (member & 1) * UW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ks = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    if (tid == 0) dead = 0;

    // Wh[unit 32 w + u0 + l16][packed column ks * 128 + 16 j + 4 kq + s] out of whbT [2][8 w][128 q][64 lane][4 s]
    float wreg[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int c = ks * 128 + 16 * j + 4 * kq + s4;
            wreg[j][s4] = a.whbT[((((size_t)(dir * 8 + w) * 128 + (c >> 3)) * 64 + ((c & 7) >> 2) * 32 + u0 + l16) << 2) + (c & 3)];
        }

    const int fu = tid & 15, frow = tid >> 4;          // one cell per lane (RH = 2: the upper half of the block idles)
    const bool fin = frow < 16 * NR;
    const int unit = w * 32 + u0 + fu;
    float dcn = 0.f;
    unsigned* ctr = a.sync + CTR_STRIDE * (1 + RH * (2 * a.tile0 + group) + half);
    constexpr int R = 16 * NR;                          // rows of this workgroup's (half) tile
    const size_t xstep = (size_t)Bp * (2 * GP);        // floats of one step in the exchange layout (= one step of dz)
    float* xbase = XCH ? a.xch + (size_t)(RH * (2 * a.tile0 + group) + half) * (S * 4 * R * 16) : nullptr;
    __syncthreads();

    auto stamp = [&](int step, int phase) {
        if (STAMPS && (tid & 63) == 0 && tid < 128 && blockIdx.x < 32 && step >= 64 && step < 72)
            a.stamps[((blockIdx.x * 8 + (step - 64)) * 2 + (tid >> 6)) * 8 + phase] = wall_clock64();
    };
    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : (T - 1 - s);
        const int tnext = dir ? t - 1 : t + 1;
        const int tp = dir ? t + 1 : t - 1;
        const bool has_prev = dir ? (t + 1 < T) : (t > 0);
        const size_t row0 = (size_t)t * Bp + b0;
        stamp(s, 0);

        const size_t row = row0 + (fin ? frow : 0);
        const float* rv = a.resv + row * (2 * 5 * HP) + dir * 5 * HP + unit;
        float dh = 0.f, ig = 0.f, jg = 0.f, fg = 0.f, og = 0.f, cc = 0.f, cp = 0.f;
        float touched[4] = {0.f, 0.f, 0.f, 0.f};
        if (fin) {
            dh = a.dhout[row * (2 * HP) + dir * HP + unit];
            ig = rv[0 * HP], jg = rv[1 * HP], fg = rv[2 * HP], og = rv[3 * HP], cc = rv[4 * HP];
            cp = has_prev ? a.resv[((size_t)tp * Bp + b0 + frow) * (2 * 5 * HP) + dir * 5 * HP + 4 * HP + unit] : 0.f;
            // touch the four dz lines this lane will store to two steps from now (see the forward kernel)
            const int s2 = XCH ? s : (s + 2 < T ? s + 2 : T - 1);
            const int t2 = dir ? s2 : (T - 1 - s2);
            const float* z2 = a.dz + ((size_t)t2 * Bp + b0 + frow) * (2 * GP) + dir * GP + w * 128 + u0 + fu;
            if (!XCH)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(touched[g]) : "v"(z2), "i"(128 * g) : "memory");
        }

        f32x4 acc[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        if (s > 0) {
            if (tid == 0 && !dead) {
                const unsigned want = (unsigned)S * (unsigned)s;
                unsigned polls = 0;
                long long t0 = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (avsi_spin_expired(polls, t0, a.spin_ticks, a.sync)) {
                        dead = 1;
                        atomicExch(a.sync, 1u);
                        break;
                    }
                }
            }
            stamp(s, 1);
            __syncthreads();
            stamp(s, 2);
            // rows l16 and 16 + l16 of the tile, columns ks * 128 + 16 j + 4 kq .. + 3
            const float* zp = a.dz + ((size_t)tnext * Bp + b0 + l16) * (2 * GP) + dir * GP + ks * 128 + 4 * kq;
            const float* zq = zp + (size_t)16 * (2 * GP);
            // (starting the MFMAs on the first loads while the rest is still landing -- staged vmcnt waits -- was slower)
            v4f a0[8], a1[8];
            if (XCH) {
                // columns ks * 128 + 16 j + 4 kq ..: block (member 2 ks + (j & 1), gate j >> 1), row l16 (and 16 + l16)
                const float* xp = xbase + (size_t)(s - 1) * xstep + (size_t)l16 * 16 + 4 * kq;
                if (a.coherent) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float* bp = xp + (((2 * ks + (j & 1)) * 4 + (j >> 1)) * R) * 16;
                        coherent_load4_issue<false>(a0[j], bp, 0);
                        if (NR == 2) coherent_load4_issue<false>(a1[j], bp, 16 * 16 * 4);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float* bp = xp + (((2 * ks + (j & 1)) * 4 + (j >> 1)) * R) * 16;
                        coherent_load4_issue<true>(a0[j], bp, 0);
                        if (NR == 2) coherent_load4_issue<true>(a1[j], bp, 16 * 16 * 4);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) agent_load4_issue(a0[j], zp, 64 * j);
                if (NR == 2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) agent_load4_issue(a1[j], zq, 64 * j);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("" : "+v"(a0[j]));
                if (NR == 2) asm volatile("" : "+v"(a1[j]));
            }
            stamp(s, 3);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float x0 = s4 == 0 ? a0[j].x : s4 == 1 ? a0[j].y : s4 == 2 ? a0[j].z : a0[j].w;
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, wreg[j][s4], acc[0], 0, 0, 0);
                    if (NR == 2) {
                        const float x1 = s4 == 0 ? a1[j].x : s4 == 1 ? a1[j].y : s4 == 2 ? a1[j].z : a1[j].w;
                        acc[NR - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, wreg[j][s4], acc[NR - 1], 0, 0, 0);
                    }
                }
        }
        // D: lane holds rows 4 kq .. 4 kq + 3 of column (unit) l16
#pragma unroll
        for (int i = 0; i < NR; ++i)
            *reinterpret_cast<float4*>(part + (ks * UW + l16) * PSTRIDE + 16 * i + 4 * kq) =
                make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        __syncthreads();
        stamp(s, 4);

        if (fin) {
            float d = dh;
#pragma unroll
            for (int k = 0; k < 8; ++k) d += part[(k * UW + fu) * PSTRIDE + frow];
            const float tc = 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * cc)) - 1.f;
            const float dc = d * og * (1.f - tc * tc) + dcn;
            dcn = dc * fg;
            float* zo = a.dz + row * (2 * GP) + dir * GP + w * 128 + u0 + fu;
            const float dzi = dc * jg * ig * (1.f - ig), dzj = dc * ig * (1.f - jg * jg);
            const float dzf = dc * cp * fg * (1.f - fg), dzo = d * tc * og * (1.f - og);
            if (XCH) {      // the exchange copy first (the only stores the publication waits for), dz beside it
                float* xo = xbase + (size_t)s * xstep + ((size_t)(member * 4) * R + frow) * 16 + fu;
                __hip_atomic_store(xo + 0 * R * 16, dzi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(xo + 1 * R * 16, dzj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(xo + 2 * R * 16, dzf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(xo + 3 * R * 16, dzo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" ::: "memory");      // issue order = program order: the counted wait below relies on it
                zo[0] = dzi, zo[32] = dzj, zo[64] = dzf, zo[96] = dzo;
            } else {
                __hip_atomic_store(zo + 0, dzi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(zo + 32, dzj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(zo + 64, dzf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(zo + 96, dzo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        stamp(s, 5);
        if (STAMPS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the stamps' own stores sit in the same queue)
        else if (XCH) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // stores complete in issue order: the four exchange stores are done
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(touched[g]));
        stamp(s, 6);
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(s, 7);
    }
    if (member == 0 && tid == 0 && !dead) reset_counter_when_done(ctr, (unsigned)S * (unsigned)T, a.spin_ticks, a.sync);
}

}  // namespace

extern "C" int avsi_blstm_rec_bwd_coop_f32(const float* dhout, const float* reserve, const float* whbT, float* dz, int T,
                                           int Bp, int split, int max_cus, void* workspace, size_t workspace_bytes,
                                           void* stream) {
    if (!dhout || !reserve || !whbT || !dz || T <= 0 || Bp <= 0 || (Bp & 31)) return AVSI_ERR_INVALID_ARG;
    if (split != 4 && split != 8 && split != 16 && split != 32) return AVSI_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < avsi_blstm_rec_fwd_coop_workspace_bytes(Bp)) return AVSI_ERR_WORKSPACE;
    if (coop_tiles_per_launch(split, max_cus) < 1) return AVSI_ERR_UNSUPPORTED;   // 2 * split workgroups do not fit max_cus
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    // no memset: the caller zeroes the workspace once, every launch leaves its counters at zero again
    // (reset_counter_when_done) and never touches the sticky status word
    // a workspace that also holds avsi_blstm_rec_bwd_coop_exchange_bytes(T, Bp) at AVSI_COOP_EXCHANGE_OFFSET switches the
    // fine splits (16, 32) to the exchange layout
    float* xch = (split >= 16 && workspace_bytes >= AVSI_COOP_EXCHANGE_OFFSET + avsi_blstm_rec_bwd_coop_exchange_bytes(T, Bp))
                     ? reinterpret_cast<float*>(static_cast<char*>(workspace) + AVSI_COOP_EXCHANGE_OFFSET) : nullptr;
    if (xch && avsi_blstm_rec_fwd_coop_workspace_bytes(Bp) > AVSI_COOP_EXCHANGE_OFFSET) return AVSI_ERR_WORKSPACE;
    if (split >= 16) {
        (void)hipFuncSetAttribute((const void*)blstm_rec_bwd_coop_fine_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_lds_bytes());
        (void)hipFuncSetAttribute((const void*)blstm_rec_bwd_coop_fine_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_lds_bytes());
        (void)hipFuncSetAttribute((const void*)blstm_rec_bwd_coop_fine_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_lds_bytes());
        (void)hipFuncSetAttribute((const void*)blstm_rec_bwd_coop_fine_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_lds_bytes());
    }
    const int tiles = Bp / 32, per = coop_tiles_per_launch(split, max_cus);
    for (int tile0 = 0; tile0 < tiles; tile0 += per) {
        const int nt = tiles - tile0 < per ? tiles - tile0 : per;
        CoopBwdArgs a{dhout, reserve, whbT, dz, (unsigned*)workspace, T, Bp, 2 * nt, tile0, xch, coop_coherent(), avsi_coop_spin_ticks(),
                      avsi_cs_stamps_buffer()};
        const int blocks = (int)avsi_ceil_div(2 * nt, AVSI_NUM_XCD) * AVSI_NUM_XCD * split;
        if (split == 32 && xch && a.stamps) {     // diagnostic instantiation (avsi_diag_cs_stamps)
            (void)hipFuncSetAttribute((const void*)blstm_rec_bwd_coop_fine_kernel<2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fine_lds_bytes());
            hipLaunchKernelGGL((blstm_rec_bwd_coop_fine_kernel<2, true, true>), dim3(blocks), dim3(512), (int)fine_lds_bytes(), st, a);
        } else if (split == 32 && xch)     // 16 unit slices x 2 row halves
            hipLaunchKernelGGL((blstm_rec_bwd_coop_fine_kernel<2, true>), dim3(blocks), dim3(512), (int)fine_lds_bytes(), st, a);
        else if (split == 32)
            hipLaunchKernelGGL((blstm_rec_bwd_coop_fine_kernel<2, false>), dim3(blocks), dim3(512), (int)fine_lds_bytes(), st, a);
        else if (split == 16 && xch)     // 96 KiB of LDS requested on purpose: one workgroup per CU (it uses 18 KiB)
            hipLaunchKernelGGL((blstm_rec_bwd_coop_fine_kernel<1, true>), dim3(blocks), dim3(512), (int)fine_lds_bytes(), st, a);
        else if (split == 16)
            hipLaunchKernelGGL((blstm_rec_bwd_coop_fine_kernel<1, false>), dim3(blocks), dim3(512), (int)fine_lds_bytes(), st, a);
        else if (split == 8)
            hipLaunchKernelGGL(blstm_rec_bwd_coop_kernel<8>, dim3(blocks), dim3(512), 0, st, a);
        else
            hipLaunchKernelGGL(blstm_rec_bwd_coop_kernel<4>, dim3(blocks), dim3(512), 0, st, a);
        const int rc = avsi_launch_status();
        if (rc != AVSI_OK) return rc;
    }
    return AVSI_OK;
}
