// Mid-batch form of the recurrent half of one bidirectional LSTM layer (same maths, layouts and packed
// operands as blstm_fwd.hip / blstm_fwd_coop.hip; reference models.py:95-115): 512 .. 2048 utterances,
// where the weight-stationary groups of blstm_fwd_coop.hip fill the chip but every workgroup still spends
// half of a step waiting for its peers (poll, coherent load of h, store drain) with the matrix pipe idle.
//
// blstm_fwd_coop.hip splits the REDUCTION over the waves of a workgroup, so the eight partial 32 x 128 tiles
// must be parked in LDS (147 KB: one workgroup per CU).  Here the waves split the COLUMNS instead:
//   - a group of 8 workgroups owns (ROWS = 16 or 32 utterances, direction); workgroup m owns the 32 hidden units
//     of slice m, wave v of it the four units 4v .. 4v + 3, i.e. 16 gate columns, over the whole reduction
//     (k = 0 .. 255).  Its piece of Wh is 64 VGPRs per lane, loaded once;
//   - v_mfma_f32_16x16x4_f32 computes z^T = Wh^T h^T: the A operand is the resident weight piece (rows = the
//     16 gate columns, ordered unit-major), the B operand h_{t-1}^T of 16 utterances.  A lane of the result
//     holds all four gates of ONE (utterance, unit) cell: no partial sums, no park, no transposition --
//     the cell is finished in the registers the MFMAs wrote;
//   - h_{t-1} of the group (ROWS x 256) is fetched once per workgroup and step with device-coherent 16-byte
//     loads, staged in LDS (double-buffered: ONE barrier per step) and read back as B fragments;
//   - LDS is 2 x ROWS x 260 floats (33 / 66 KB) and a wave needs < 128 registers, so TWO workgroups share a
//     CU: while one polls, loads and stores, the other's MFMAs run;
//   - exchange protocol as in blstm_fwd_coop.hip (device-coherent stores of h into hout, a per-group step
//     counter, bounded spin); the new h leaves through the idle LDS buffer in whole 128-byte rows, stored and
//     published by ONE wave while the others already prefetch the next step's input projection.
// The launch must be wholly resident; the host sizes it from the occupancy the runtime reports.
#include "avsi_common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int HP = 256, GP = 4 * HP;
constexpr int HPITCH = HP + 4;                       // LDS pitch of one utterance's h (conflict-free b128 rows)
constexpr int MEMBERS = 8;
constexpr int CTR_STRIDE = 64;                       // one 256-byte line per counter: 8 pollers each, spread over the channels

struct CsArgs {
    const float* xproj;
    const float* whp;
    float* hout;
    float* resv;
    unsigned* sync;    // [0] status, [CTR_STRIDE * (1 + group)] step counters
    int T, Bp, ngroups;
    int group0;        // first (row tile, direction) group of this launch
    unsigned long long* stamps;   // diagnostics (STAMPS): [block < 32][step 64 .. 71][wave 0 / 1][8 phases] wall clock
    long long spin_ticks;         // bound of every wait for a peer, 100 MHz ticks (avsi_coop_spin_ticks)
};

__device__ __forceinline__ float sigmoidf_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_fast(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

// device-coherent 16-byte load (see blstm_fwd_coop.hip): issue now, close the batch with one s_waitcnt
// (Plain, sc0-only and sc1-only loads were all bit-correct here and none was faster.)
__device__ __forceinline__ void coherent_load4_issue(v4f& dst, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(dst) : "v"(p) : "memory");
}

// The s_nop is part of the instruction: a VALU write to the data registers of a > 64-bit store needs two wait
// states behind it, and the compiler's hazard recogniser does not see into inline asm (without it the address
// arithmetic of the NEXT store landed in half of the lanes' data).
__device__ __forceinline__ void coherent_store4(float* p, v4f v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int RT, bool SAVE, bool STAMPS>
__global__ __launch_bounds__(512, 4) void blstm_rec_fwd_cs_kernel(const CsArgs a) {
    constexpr int ROWS = 16 * RT;
    constexpr int SPITCH = 36;                           // staging pitch of one utterance's 32 units
    constexpr int SARR = ROWS * SPITCH;                  // one staged array
    static_assert((SAVE ? 6 : 1) * SARR <= ROWS * HPITCH, "the idle h buffer doubles as the staging area");
    __shared__ __attribute__((aligned(16))) float hs[2][ROWS * HPITCH];
    __shared__ int wg_dead;

    // block -> (group, member): members of a group are 8 block ids apart (same XCD: a locality hint only)
    const int xcd = blockIdx.x % AVSI_NUM_XCD, kk = blockIdx.x / AVSI_NUM_XCD;
    const int member = kk % MEMBERS;
    const int lgroup = (kk / MEMBERS) * AVSI_NUM_XCD + xcd;
    if (lgroup >= a.ngroups) return;
    // behind a launch that gave up a bounded wait every result is void (sticky status word): leave at once instead of
    // waiting out the time bound on step counters that launch may have left behind
    if (avsi_launch_is_void(a.sync)) return;
    const int group = a.group0 + lgroup;
    const int dir = group & 1;
    const int b0 = (group >> 1) * ROWS;
    const int T = a.T, Bp = a.Bp;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;

    // ---- this wave's piece of Wh, once.  MFMA (J, e), J = 0 .. 15, e = 0 .. 3, consumes k = 16 J + 4 kq + e;
    //      the A lane (row n = l16 -> unit 4 v + n / 4, gate n % 4; k index kq) takes it from the packed
    //      whp [2][8 w][32 q][4 g][64 lane][4 s] = Wh[k = 8 q + 4 (lane / 32) + s][gate g, unit 32 w + lane % 32]
    float4 wreg[16];
    {
        const int un = 4 * v + (l16 >> 2), g = l16 & 3;
        const float4* wp = reinterpret_cast<const float4*>(a.whp) + (size_t)(dir * 8 + member) * (32 * 4 * 64);
#pragma unroll
        for (int J = 0; J < 16; ++J) wreg[J] = wp[((2 * J + (kq >> 1)) * 4 + g) * 64 + (kq & 1) * 32 + un];
    }

    // ---- this lane's cells: utterance l16 of every 16-row tile, unit 4 v + kq of the slice
    float cstate[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) cstate[r] = 0.f;

    unsigned* ctr = a.sync + CTR_STRIDE * (1 + group);
    if (tid == 0) wg_dead = 0;
    __syncthreads();
    auto stamp = [&](int step, int phase) {
        if (STAMPS && lane == 0 && v < 2 && blockIdx.x < 32 && step >= 64 && step < 72)
            a.stamps[((blockIdx.x * 8 + (step - 64)) * 2 + v) * 8 + phase] = wall_clock64();
    };

    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        const int tprev = dir ? t + 1 : t - 1;
        const size_t row0 = (size_t)t * Bp + b0;
        float* hbuf = hs[step & 1];
        float* stage = hs[(step + 1) & 1];               // idle until the next step's h arrives
        stamp(step, 0);

        // hoisted input projection of this lane's cells (packed [slice][gate][32 units])
        float xz[RT][4];
        auto load_xz = [&]() {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    xz[r][g] = a.xproj[(row0 + 16 * r + l16) * (2 * GP) + dir * GP + member * 128 + g * 32 + 4 * v + kq];
        };
        if (step == 0) load_xz();

        v4f acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = v4f{0.f, 0.f, 0.f, 0.f};

        if (step > 0) {
            // ---- wait until all members have published h of the previous step.  (Every wave polling and
            //      publishing by itself -- no barriers -- was 7 x slower: 64 increments and 4096 pollers per
            //      step on one address.)
            if (tid == 0 && !wg_dead) {
                const unsigned want = (unsigned)MEMBERS * (unsigned)step;
                unsigned polls = 0;
                long long t0 = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (avsi_spin_expired(polls, t0, a.spin_ticks, a.sync)) {
                        wg_dead = 1;
                        atomicExch(a.sync, 1u);
                        break;
                    }
                }
            }
            stamp(step, 1);
            AVSI_LDS_BARRIER();       // also: every wave has read its share of the staging area (= hbuf) back
            stamp(step, 2);
            // ---- h_{t-1} of the group -> LDS: float4 f = tid + 512 i is (row f / 64, columns 4 (f % 64) ..)
            {
                v4f hv[2 * RT];
                const float* hp = a.hout + ((size_t)tprev * Bp + b0) * (2 * HP) + dir * HP;
#pragma unroll
                for (int i = 0; i < 2 * RT; ++i) {
                    const int f = tid + 512 * i;
                    coherent_load4_issue(hv[i], hp + (size_t)(f >> 6) * (2 * HP) + 4 * (f & 63));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                stamp(step, 3);
#pragma unroll
                for (int i = 0; i < 2 * RT; ++i) {
                    asm volatile("" : "+v"(hv[i]));
                    const int f = tid + 512 * i;
                    *reinterpret_cast<v4f*>(hbuf + (f >> 6) * HPITCH + 4 * (f & 63)) = hv[i];
                }
            }
            AVSI_LDS_BARRIER();
            stamp(step, 4);
            // requested here, behind the MFMAs: in front of the exchange (at the top of the step) these 16-byte-granular
            // loads sat in the memory pipeline ahead of the h stores, whose acknowledgement then took 4.5 us instead of 1.2
            load_xz();

            // ---- z^T (16 gate columns x 16 utterances per tile) += Wh^T piece . h^T
            const float* hb = hbuf + l16 * HPITCH + 4 * kq;
            v4f hf[2][RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) hf[0][r] = *reinterpret_cast<const v4f*>(hb + 16 * r * HPITCH);
#pragma unroll
            for (int J = 0; J < 16; ++J) {
                if (J + 1 < 16)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        hf[(J + 1) & 1][r] = *reinterpret_cast<const v4f*>(hb + 16 * r * HPITCH + 16 * (J + 1));
                const float4 w = wreg[J];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float wv = e == 0 ? w.x : e == 1 ? w.y : e == 2 ? w.z : w.w;
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, hf[J & 1][r][e], acc[r], 0, 0, 0);
                }
            }
        }

        // ---- finish this lane's cells: acc[r] = (i, j, f, o) pre-activations of (utterance 16 r + l16, unit 4 v + kq);
        //      results are staged [array][utterance][32 units] so that they leave in 128-byte rows (a lane's own
        //      4-byte stores would be 16-byte fragments: write-through traffic in eighths of a line)
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const float ig = sigmoidf_fast(acc[r][0] + xz[r][0]), jg = tanhf_fast(acc[r][1] + xz[r][1]);
            const float fg = sigmoidf_fast(acc[r][2] + xz[r][2]), og = sigmoidf_fast(acc[r][3] + xz[r][3]);
            const float cn = fg * cstate[r] + ig * jg;
            cstate[r] = cn;
            float* sp = stage + (16 * r + l16) * SPITCH + 4 * v + kq;
            sp[0] = og * tanhf_fast(cn);
            if (SAVE) sp[1 * SARR] = ig, sp[2 * SARR] = jg, sp[3 * SARR] = fg, sp[4 * SARR] = og, sp[5 * SARR] = cn;
        }
        stamp(step, 5);
        AVSI_LDS_BARRIER();
        stamp(step, 6);

        // ---- wave 0 stores h (device-coherent) and publishes; waves 1 .. 5 store the reserve arrays
        if (v == 0 || (SAVE && v <= 5)) {
            v4f sv[ROWS / 8];
#pragma unroll
            for (int i = 0; i < ROWS / 8; ++i) {
                const int f = lane + 64 * i;             // (utterance f / 8, units 4 (f % 8) ..)
                sv[i] = *reinterpret_cast<const v4f*>(stage + v * SARR + (f >> 3) * SPITCH + 4 * (f & 7));
            }
            if (v == 0) {
#pragma unroll
                for (int i = 0; i < ROWS / 8; ++i) {
                    const int f = lane + 64 * i;
                    coherent_store4(a.hout + (row0 + (f >> 3)) * (2 * HP) + dir * HP + member * 32 + 4 * (f & 7), sv[i]);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stamp(step, 7);
            } else {
#pragma unroll
                for (int i = 0; i < ROWS / 8; ++i) {
                    const int f = lane + 64 * i;
                    *reinterpret_cast<v4f*>(a.resv + (row0 + (f >> 3)) * (2 * 5 * HP) + dir * 5 * HP + (v - 1) * HP + member * 32 +
                                            4 * (f & 7)) = sv[i];
                }
            }
        }
    }
    // leave the counter at zero for the next launch on this workspace (see blstm_fwd_coop.hip)
    if (member == 0 && tid == 0 && !wg_dead) {
        unsigned polls = 0;
        long long t0 = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)MEMBERS * (unsigned)T) {
            __builtin_amdgcn_s_sleep(2);
            if (avsi_spin_expired(polls, t0, a.spin_ticks, a.sync)) return;
        }
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

unsigned long long* g_cs_stamps = nullptr;     // avsi_diag_cs_stamps()

template <int RT, bool SAVE>
int cs_blocks_per_cu() {
    static int cached = -1;
    if (cached < 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)blstm_rec_fwd_cs_kernel<RT, SAVE, false>, 512, 0) != hipSuccess)
            n = 1;
        cached = n < 1 ? 1 : (n > 2 ? 2 : n);
    }
    return cached;
}

template <int RT, bool SAVE>
int launch_cs(const float* xproj, const float* whp, float* hout, float* reserve, int T, int Bp, int first_row, int rows,
              int max_cus, unsigned* sync, hipStream_t st) {
    const int cus = (max_cus <= 0 || max_cus > AVSI_NUM_CU) ? AVSI_NUM_CU : max_cus;
    // whole groups, in multiples of the XCD count when there are that many (members sit 8 block ids apart)
    int per = cus * cs_blocks_per_cu<RT, SAVE>() / MEMBERS;
    if (per >= AVSI_NUM_XCD) per = per / AVSI_NUM_XCD * AVSI_NUM_XCD;
    if (per < 1) return AVSI_ERR_UNSUPPORTED;
    // groups 2 * tile + direction of the utterance tiles [first_row, first_row + rows) / (16 RT)
    const int gbeg = 2 * (first_row / (16 * RT)), groups = gbeg + 2 * (rows / (16 * RT));
    for (int g0 = gbeg; g0 < groups; g0 += per) {
        const int ng = groups - g0 < per ? groups - g0 : per;
        CsArgs a{xproj, whp, hout, reserve, sync, T, Bp, ng, g0, g_cs_stamps, avsi_coop_spin_ticks()};
        const int blocks = (int)avsi_ceil_div(ng, AVSI_NUM_XCD) * AVSI_NUM_XCD * MEMBERS;
        if (g_cs_stamps)
            hipLaunchKernelGGL((blstm_rec_fwd_cs_kernel<RT, SAVE, true>), dim3(blocks), dim3(512), 0, st, a);
        else
            hipLaunchKernelGGL((blstm_rec_fwd_cs_kernel<RT, SAVE, false>), dim3(blocks), dim3(512), 0, st, a);
        const int rc = avsi_launch_status();
        if (rc != AVSI_OK) return rc;
    }
    return AVSI_OK;
}

}  // namespace

// (utterance tile, direction) groups one launch holds: 8 workgroups each, as many per CU as the runtime reports (<= 2)
extern "C" int avsi_blstm_rec_fwd_cs_groups_per_launch(int rows_per_group, int with_reserve, int max_cus) {
    const int cus = (max_cus <= 0 || max_cus > AVSI_NUM_CU) ? AVSI_NUM_CU : max_cus;
    int per_cu;
    if (rows_per_group == 16)
        per_cu = with_reserve ? cs_blocks_per_cu<1, true>() : cs_blocks_per_cu<1, false>();
    else if (rows_per_group == 32)
        per_cu = with_reserve ? cs_blocks_per_cu<2, true>() : cs_blocks_per_cu<2, false>();
    else
        return 0;
    return cus * per_cu / MEMBERS;
}

// Diagnostic: while `buffer` (32 * 8 * 2 * 8 uint64, device memory) is set, launches record the wall clock of eight
// phases of steps 64 .. 71 for waves 0 and 1 of the first 32 workgroups (tools/rec_cs_stamps.py); null switches it off.
// internal (blstm_fwd_coop.hip: the 32-way kernel's diagnostic instantiation records into the same buffer)
unsigned long long* avsi_cs_stamps_buffer() { return g_cs_stamps; }

extern "C" int avsi_diag_cs_stamps(void* buffer) {
    g_cs_stamps = (unsigned long long*)buffer;
    return AVSI_OK;
}

extern "C" size_t avsi_blstm_rec_fwd_cs_workspace_bytes(int Bp) {
    return (size_t)CTR_STRIDE * (1 + 2 * (Bp > 0 ? (Bp + 15) / 16 : 0)) * sizeof(unsigned);
}

extern "C" int avsi_blstm_rec_fwd_cs_rows_f32(const float* xproj, const float* whp, float* hout, float* reserve, int T, int Bp,
                                              int rows_per_group, int first_row, int rows, int max_cus, void* workspace,
                                              size_t workspace_bytes, void* stream) {
    if (!xproj || !whp || !hout || T <= 0 || Bp <= 0 || (Bp & 31)) return AVSI_ERR_INVALID_ARG;
    if (rows_per_group != 16 && rows_per_group != 32) return AVSI_ERR_INVALID_ARG;
    if (first_row < 0 || rows <= 0 || first_row % rows_per_group || rows % rows_per_group || first_row + rows > Bp)
        return AVSI_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < avsi_blstm_rec_fwd_cs_workspace_bytes(Bp)) return AVSI_ERR_WORKSPACE;
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    // no memset: the caller zeroes the workspace once, every launch leaves its counters at zero again and never
    // touches the sticky status word
    unsigned* sync = (unsigned*)workspace;
    if (rows_per_group == 16)
        return reserve ? launch_cs<1, true>(xproj, whp, hout, reserve, T, Bp, first_row, rows, max_cus, sync, st)
                       : launch_cs<1, false>(xproj, whp, hout, reserve, T, Bp, first_row, rows, max_cus, sync, st);
    return reserve ? launch_cs<2, true>(xproj, whp, hout, reserve, T, Bp, first_row, rows, max_cus, sync, st)
                   : launch_cs<2, false>(xproj, whp, hout, reserve, T, Bp, first_row, rows, max_cus, sync, st);
}

extern "C" int avsi_blstm_rec_fwd_cs_f32(const float* xproj, const float* whp, float* hout, float* reserve, int T, int Bp,
                                         int rows_per_group, int max_cus, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    return avsi_blstm_rec_fwd_cs_rows_f32(xproj, whp, hout, reserve, T, Bp, rows_per_group, 0, Bp, max_cus, workspace,
                                          workspace_bytes, stream);
}
