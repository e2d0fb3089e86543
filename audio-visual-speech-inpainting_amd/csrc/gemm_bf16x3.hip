// EXPLORATORY (never the default, never the headline number): C = A . B + bias with every fp32 operand split into
// two bf16 values (hi = round(a), lo = round(a - hi): 16+ significant bits) and the product taken as
// hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation ("bf16x3").  It answers one question:
// what would the 1e-3 RMS tolerance of BASELINE.json buy beyond the 157 TFLOP/s fp32-MFMA ceiling that the shipped
// path (exact fp32 fma chains, `dtype: "f32"`) lives under?  Used for the layer input projections x . Wx + b of the
// stacked BLSTM (reference models.py:95-115) when a model is built with config['precision'] = 'bf16x3'
// (`bench.py --precision bf16x3`, which prints the log-mel RMS against the CPU oracle next to the rate).
//
// 128 x 128 block tile, 4 waves (2 x 2) of 64 x 64 (2 x 2 MFMA tiles), k-tile 32.  A is split ONCE per element, on
// the way from global memory to LDS (register-staged loads, v_cvt_pk_bf16_f32), and read back as ready bf16 fragments
// with one ds_read_b128 each ([row][k], row pitch 40 bf16 = 80 B: conflict-free).  B -- the weights, the same for
// every row block and every call -- is split and laid out in MFMA fragment order beforehand
// (avsi_pack_bf16x3_b: [column tile][k16 step][lane][8] for hi and for lo), so a lane fetches its fragment with one
// 16-byte load from L2 and B never touches LDS.  (The first version staged B through LDS as well, transposing it with
// 4-byte writes at a 4-column lane stride: 8-way bank conflicts, LDS-bound at 210 TFLOP/s fp32-equivalent.)
#include "avsi_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, PITCH = 40;     // PITCH in bf16 elements
constexpr int PLANE = 128 * PITCH;                          // bf16 elements per operand plane

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {      // (lo half = a, hi half = b), round to nearest even
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = __builtin_convertvector((f32x2){a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf16_lo_as_float(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi_as_float(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// hi / lo split of a pair: returns the packed hi pair, stores the packed lo pair
__device__ __forceinline__ unsigned split2(float a, float b, unsigned& lo) {
    const unsigned hi = pack_bf16(a, b);
    lo = pack_bf16(a - bf16_lo_as_float(hi), b - bf16_hi_as_float(hi));
    return hi;
}

// Out-of-range operand pieces are READ FROM HERE instead of being zeroed after the load: a `ok ? load : 0` select lets
// the compiler predicate the load, and a load in a predicated block is closed with s_waitcnt vmcnt(0) -- the whole
// prefetch then sits in front of the MFMAs instead of under them (DESIGN §4.1 has the same story for the front end).
__device__ float g_zero_page[512];

struct G3Args {
    const float* A;
    const __bf16* Bh;      // packed hi plane
    const __bf16* Bl;      // packed lo plane
    const float* bias;
    float* C;
    int M, N, K;
    int64_t lda, ldc;
    int m_blocks, n_blocks, ksteps;       // ksteps = k16 steps of the packed B (K rounded up to 32, / 16)
};

// B [K][N] fp32 -> fragment order of v_mfma_f32_32x32x16_bf16: element e of lane l of (column tile ct, k16 step ks) is
// B[16 ks + 8 (l / 32) + e][32 ct + l % 32]; rows past K are zero
__global__ void pack_b_kernel(const float* __restrict__ B, int64_t ldb, int K, int N, int ksteps, __bf16* __restrict__ hi,
                              __bf16* __restrict__ lo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one (ct, ks, lane)
    const int64_t total = (int64_t)(N / 32) * ksteps * 64;
    if (idx >= total) return;
    const int lane = (int)(idx % 64);
    const int ks = (int)((idx / 64) % ksteps), ct = (int)(idx / 64 / ksteps);
    const int n = 32 * ct + (lane & 31), kb = 16 * ks + 8 * (lane >> 5);
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const float x0 = kb + e < K ? B[(int64_t)(kb + e) * ldb + n] : 0.f;
        const float x1 = kb + e + 1 < K ? B[(int64_t)(kb + e + 1) * ldb + n] : 0.f;
        const unsigned hp = pack_bf16(x0, x1);
        h[e / 2] = hp;
        l[e / 2] = pack_bf16(x0 - bf16_lo_as_float(hp), x1 - bf16_hi_as_float(hp));
    }
    *reinterpret_cast<uint4*>(hi + idx * 8) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(lo + idx * 8) = make_uint4(l[0], l[1], l[2], l[3]);
}

__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const G3Args g) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * PLANE];      // A hi | A lo  (20 KB)
    __bf16* a_hi = lds;
    __bf16* a_lo = lds + PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // column blocks of one row block are neighbours in the grid: the A panel is shared through L2
    const int mb = blockIdx.x / g.n_blocks, nb = blockIdx.x % g.n_blocks;
    const int m0 = mb * BM, n0 = nb * BN;

    // staging map of A: 4 float4 per thread (row = tid / 8 + 32 i, k = 4 (tid % 8))
    const int a_row = tid >> 3, a_k = (tid & 7) * 4;
    float4 ra[4];
    // B fragments of this wave's two column tiles: (ct, ks, lane) -> 8 bf16
    const int ct0 = (n0 + wn * 64) / 32;
    const uint4* bh_p = reinterpret_cast<const uint4*>(g.Bh) + (int64_t)ct0 * g.ksteps * 64 + lane;
    const uint4* bl_p = reinterpret_cast<const uint4*>(g.Bl) + (int64_t)ct0 * g.ksteps * 64 + lane;
    const int64_t ct_stride = (int64_t)g.ksteps * 64;
    uint4 rbh[2][2], rbl[2][2];     // [k16 step of the tile][column tile]

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // rows past M read row M - 1 (their results are never stored); k past K reads zeros (K % 4 == 0: a float4
            // is inside or outside as a whole)
            const int row = min(m0 + a_row + 32 * i, g.M - 1), k = k0 + a_k;
            const float* src = k < g.K ? g.A + (int64_t)row * g.lda + k : g_zero_page + a_k;
            ra[i] = *reinterpret_cast<const float4*>(src);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t o = (int64_t)(k0 / 16 + st) * 64 + t * ct_stride;
                rbh[st][t] = bh_p[o];
                rbl[st][t] = bl_p[o];
            }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned l0, l1;
            const unsigned h0 = split2(ra[i].x, ra[i].y, l0), h1 = split2(ra[i].z, ra[i].w, l1);
            const int off = (a_row + 32 * i) * PITCH + a_k;
            *reinterpret_cast<uint2*>(a_hi + off) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(a_lo + off) = make_uint2(l0, l1);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int frag_k = 8 * (lane >> 5);                    // this lane's 8 consecutive k of a k16 step
    const int arow0 = wm * 64 + (lane & 31);

    load_tile(0);
    for (int k0 = 0; k0 < g.ksteps * 16; k0 += BK) {
        store_tile();
        bf16x8 bhc[2][2], blc[2][2];                       // this tile's B fragments (the prefetch registers are reloaded)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bhc[st][t] = __builtin_bit_cast(bf16x8, rbh[st][t]);
                blc[st][t] = __builtin_bit_cast(bf16x8, rbl[st][t]);
            }
        AVSI_LDS_BARRIER();
        if (k0 + BK < g.ksteps * 16) load_tile(k0 + BK);   // in flight under this tile's MFMAs
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ao = (arow0 + 32 * t) * PITCH + 16 * s + frag_k;
                ah[t] = *reinterpret_cast<const bf16x8*>(a_hi + ao);
                al[t] = *reinterpret_cast<const bf16x8*>(a_lo + ao);
                bh[t] = bhc[s][t];
                bl[t] = blc[s][t];
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    // small terms first: hi.lo and lo.hi are ~2^-8 of hi.hi
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                }
        }
        AVSI_LDS_BARRIER();
    }

    // epilogue: D[row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col lane & 31] of every 32 x 32 tile.  Whole tiles take
    // straight-line stores: under a per-lane row test every store sits in its own block and is closed by a wait.
    const bool whole = m0 + BM <= g.M;       // workgroup-uniform
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + wn * 64 + nt * 32 + (lane & 31);
        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row0 = m0 + wm * 64 + mt * 32 + 4 * (lane >> 5);
            float* cp = g.C + (int64_t)row0 * g.ldc + col;
            if (whole) {
#pragma unroll
                for (int r = 0; r < 16; ++r) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = acc[mt][nt][r] + bv;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (row0 + (r & 3) + 8 * (r >> 2) < g.M) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = acc[mt][nt][r] + bv;
            }
        }
    }
}

}  // namespace

extern "C" size_t avsi_pack_bf16x3_b_bytes(int K, int N) {
    if (K <= 0 || N <= 0 || (N % 32)) return 0;
    const int ksteps = (int)avsi_round_up(K, BK) / 16;
    return (size_t)2 * (size_t)(N / 32) * ksteps * 64 * 8 * sizeof(__bf16);      // hi plane, then lo plane
}

extern "C" int avsi_pack_bf16x3_b(const float* B, int64_t ldb, int K, int N, void* packed, size_t packed_bytes, void* stream) {
    if (!B || !packed || K <= 0 || N <= 0 || ldb < N) return AVSI_ERR_INVALID_ARG;
    if (N % 32) return AVSI_ERR_UNSUPPORTED;
    if (packed_bytes < avsi_pack_bf16x3_b_bytes(K, N) || (reinterpret_cast<uintptr_t>(packed) & 15)) return AVSI_ERR_WORKSPACE;
    const int ksteps = (int)avsi_round_up(K, BK) / 16;
    const int64_t total = (int64_t)(N / 32) * ksteps * 64;
    __bf16* hi = static_cast<__bf16*>(packed);
    avsi_clear_error();
    hipLaunchKernelGGL(pack_b_kernel, dim3((unsigned)avsi_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, B, ldb, K, N,
                       ksteps, hi, hi + total * 8);
    return avsi_launch_status();
}

extern "C" int avsi_gemm_bf16x3_f32(int M, int N, int K, const float* A, int64_t lda, const void* packed_b, const float* bias,
                                    float* C, int64_t ldc, void* stream) {
    if (!A || !packed_b || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldc < N) return AVSI_ERR_INVALID_ARG;
    if ((N % BN) || (K & 3) || (lda & 3) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(packed_b)) & 15))
        return AVSI_ERR_UNSUPPORTED;
    const int ksteps = (int)avsi_round_up(K, BK) / 16;
    const __bf16* hi = static_cast<const __bf16*>(packed_b);
    G3Args g{A, hi, hi + (int64_t)(N / 32) * ksteps * 64 * 8, bias, C, M, N, K, lda, ldc, (int)avsi_ceil_div(M, BM), N / BN, ksteps};
    if ((int64_t)g.m_blocks * g.n_blocks > INT32_MAX) return AVSI_ERR_UNSUPPORTED;
    avsi_clear_error();
    hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3(g.m_blocks * g.n_blocks), dim3(256), 0, (hipStream_t)stream, g);
    return avsi_launch_status();
}
