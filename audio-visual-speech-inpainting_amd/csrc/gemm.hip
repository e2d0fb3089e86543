// fp32 GEMM on the gfx950 matrix cores: C = alpha * op(A) . op(B) (+ bias[n]) (+ beta * C)
//
// Used for the time-batched halves of the BLSTM (reference models.py:95-115): the hoisted input
// projections X . Wx of every layer, the output projection (models.py:119-122) and, in training,
// dX = dZ . W^T and dW = X^T . dZ.
//
// v_mfma_f32_32x32x2_f32 is exact fp32 (one rounding per product, k-ordered fma chain), so parity
// with the fp32 reference graph needs no precision argument.  It issues once per 64 cycles per
// SIMD, which leaves LDS and global loads far off the critical path; the design therefore
// optimises for "never stall the MFMA pipe":
//   - 128 x 128 x 32 block tile, 4 waves as 2 x 2, each wave 2 x 2 MFMA tiles (64 acc VGPRs),
//     two workgroups per CU so a wave waiting at the k-tile barrier is covered by its SIMD mate;
//   - register-staged double buffering: the next k-tile's 16-byte global loads are issued before
//     the current tile's 64 MFMAs and written to the other LDS stage after them;
//   - operand tiles sit in LDS in the layout they have in memory (no transposing stores):
//       "row" tiles [x][k] (k contiguous, stride 36) are read with ONE ds_read_b128 per 4 k-steps
//       using a k-permutation inside each 8-wide k group (lane half h owns k = 8q+4h..8q+4h+3);
//       "col" tiles [k][x] (x contiguous, stride 132) are read with ds_read_b32, lanes along x;
//     both are bank-conflict free, and both sides use the same permutation so products match;
//   - blockIdx -> tile mapping is XCD-aware: the N-blocks that share an A row panel get
//     consecutive ids on ONE XCD, so the panel is fetched from HBM once and re-served by that L2.
#include "avsi_common.h"
#include <atomic>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BN = 128;
// Operand tile of X rows/cols (128 or 256) by BK (16 or 32) reduction steps.
//   "row" tile [x][k]: k contiguous, stride BK + 4 floats (conflict-free ds_read_b128)
//   "col" tile [k][x]: x contiguous, stride X + 4 floats (lanes along x, ds_read_b32)
template <int BK, int X> struct Tile {
    static constexpr int ROW_STRIDE = BK + 4;
    static constexpr int COL_STRIDE = X + 4;
    static constexpr int ROW_TILE = X * ROW_STRIDE;
    static constexpr int COL_TILE = BK * COL_STRIDE;
    static constexpr int NLD = X * BK / 4 / 256;  // float4 loads per thread per operand tile
};

template <int BK, int X> using RegTile = float4[Tile<BK, X>::NLD];  // one thread's share of an operand tile
template <int BK, int X> using OffTile = int[Tile<BK, X>::NLD];     // and its float4 offsets from the tile origin

// Tile counters of the persistent wide-tile launches: 256 launch slots of 8 counters (+ padding to a 64-byte line), taken in turn
// and zeroed on the launch's stream right in front of it.  A module global: nothing is allocated behind the caller's back; a slot
// comes round again 256 persistent launches later (a model step has four).
__device__ unsigned avsi_gemm_tile_ctr[256 * 16];

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* row_scale;
    int row_map_bp, row_map_t, row_map_b;
    int M, N, K;
    int64_t lda, ldb, ldc;
    float alpha, beta;
    int m_blocks, n_blocks;
    int k_split_len;       // reduction length handled per blockIdx.z (multiple of BK)
    int64_t c_split_stride;  // elements between per-split partial outputs (0 = no split)
    int bnt;               // N tile width chosen by the host (128, or 64 for A . B with a narrow last tile)
    // N = 256 n + 1 on the 256-wide tile (the 257-bin projection): column N - 1 is not a tile of its own -- the workgroups of the
    // last column block take it on the VALU, as a dot product of the A rows they have staged in LDS anyway (gemm_dma_kernel<.., TAIL>)
    int tail_col;          // that column's index, or -1
    // PERS: eight zeroed counters (one per XCD class of workgroups): the next tile index of that class.  A workgroup TAKES its tiles
    // in order instead of walking a fixed stride: workgroups drift apart over 250 tiles, and with a fixed stride the tiles in flight
    // then spread over many row blocks -- the column tiles of a row block stop sharing its A panel in L2 (measured: 26.6 GB
    // fetched per K = 512 launch against 16.1 GB with one workgroup per tile; taken in order: 23.2 GB and 2 % faster than either)
    unsigned* tile_ctr;
    int diag;              // AVSI_GEMM_DIAG (timing experiments): 1 = the wide tile's fast epilogue stores nothing (results WRONG), 2 = it stores non-temporally, 8 = 16-byte stores behind a quad transpose (measured slower)
    int n_group;           // N-blocks per column group of the block -> tile order (see tile_of_block)
    // up to two 16-deep k-tiles that END in zero padding (avsi_gemm_epilogue::k_zero) and the number of their eight MFMA
    // steps that multiply anything: step 4 q + s multiplies k = 8 q + s and 8 q + 4 + s, so a tile with kv leading real k
    // needs its first min(kv, 4) steps, or 4 + min(kv - 8, 4) from kv = 9 on
    int sp_kt[2], sp_steps[2];
    // implicit-GEMM convolution (gemm_dma_kernel<.., CONV = true>): A is never materialised, its rows are
    // gathered from one or two NHWC activations (tf.nn.conv2d SAME / stride 1 over concat(src0, up2x(src1)))
    const float* conv_s0;
    const float* conv_s1;
    const float* conv_zeros;   // >= 64 bytes of zeros: the source of out-of-image taps
    int cH, cW, cC0, cld0, cC1, cld1, ck;
    // CONV only: per (M-block, wave row) partial column sums of the OUTPUT, [parts][2][N] (sum, sum of squares) -- the batch-norm
    // statistics of a convolution taken where its accumulators are, instead of a pass over its output (avsi_conv2d_bn_f32)
    float* stats;
};

// Block -> tile map.  Two things are arranged here:
//  * XCD awareness: hardware deals workgroups round-robin to the 8 XCDs (blocks b and b+8 share
//    an XCD and its L2), so the linear tile order is cut into 8 contiguous ranges, one per XCD;
//  * column groups: the linear order is (column group, M-block, N-block inside the group).  While
//    an XCD walks down the M-blocks of one group, the group's slice of B (K x n_group*128 floats,
//    sized by the host to about half an L2) stays resident and every A panel is read once by the
//    n_group workgroups that run side by side.  With the plain (M-block, N-block) order the whole
//    of B (4 MiB for the 512 x 2048 layer kernel = one L2) was evicted by the A / C streams and
//    re-fetched through the fabric by almost every M-block: 39.5 GB per launch against 4.2 GB of A.
__device__ __forceinline__ void tile_of_block(const GemmArgs& g, int bid, int& bm, int& bn) {
    const int nblk = g.m_blocks * g.n_blocks;
    const int q = nblk / AVSI_NUM_XCD, r = nblk % AVSI_NUM_XCD;
    const int xcd = bid % AVSI_NUM_XCD, idx = bid / AVSI_NUM_XCD;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = g.m_blocks * g.n_group;
    const int grp = lin / per_group, rem = lin - grp * per_group;
    const int width = min(g.n_group, g.n_blocks - grp * g.n_group);
    bm = rem / width;
    bn = grp * g.n_group + (rem - bm * width);
}

// Global -> registers for one operand tile.  ROWK: memory is [x][k] (k contiguous), else [k][x].
template <bool ROWK, int BK, int XT>
__device__ __forceinline__ void load_tile(RegTile<BK, XT>& r, const float* __restrict__ base, int64_t ld, int x0, int k0,
                                          int X, int Kend, int tid) {
    constexpr int KQ = BK / 4;           // float4 per row of a "row" tile
    constexpr int RPP = 256 / KQ;        // rows covered per pass
    constexpr int XQ = XT / 4;           // float4 per k-row of a "col" tile
    constexpr int KPP = 256 / XQ;        // k-rows covered per pass
#pragma unroll
    for (int p = 0; p < Tile<BK, XT>::NLD; ++p) {
        int x, k;
        if (ROWK) {
            x = x0 + p * RPP + tid / KQ;
            k = k0 + 4 * (tid % KQ);
        } else {
            k = k0 + p * KPP + tid / XQ;
            x = x0 + 4 * (tid % XQ);
        }
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (ROWK) {
            if (x < X && k < Kend) v = *reinterpret_cast<const float4*>(base + (int64_t)x * ld + k);
        } else {
            if (k < Kend && x < X) v = *reinterpret_cast<const float4*>(base + (int64_t)k * ld + x);
        }
        r[p] = v;
    }
}

// Interior fast path.  The per-thread part of every tile address is the same for all k-tiles, so
// it is computed ONCE (float4 index relative to the tile origin, rows clamped into the matrix so
// no guard is needed: out-of-range rows only feed outputs that are never stored); per k-tile the
// tile origin is one wave-uniform pointer, made opaque so each load is
// "global_load_dwordx4 v, v_off, s[base]" with no per-load VALU address math and no exec branches
// (those cost ~10 % of the MFMA rate in the guarded path: 131 -> 145 TFLOP/s at K = 4096).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f* gptr4;
__device__ __forceinline__ gptr4 opaque_base(const float* p) {
    gptr4 g = (gptr4)(const void*)p;
    asm volatile("" : "+s"(g));
    return g;
}

template <bool ROWK, int BK, int XT>
__device__ __forceinline__ void tile_offsets(OffTile<BK, XT>& off, int64_t ld, int x0, int X, int tid) {
    constexpr int KQ = BK / 4, RPP = 256 / KQ, XQ = XT / 4, KPP = 256 / XQ;
#pragma unroll
    for (int p = 0; p < Tile<BK, XT>::NLD; ++p) {
        if (ROWK) {
            const int x = min(x0 + p * RPP + tid / KQ, X - 1) - x0;            // row, clamped into the matrix
            off[p] = (int)((x * ld) >> 2) + (tid % KQ);
        } else {
            const int xq = min(x0 + 4 * (tid % XQ), ((X + 3) & ~3) - 4) - x0;   // column group, clamped into the padded row
            off[p] = (int)(((p * KPP + tid / XQ) * ld + xq) >> 2);
        }
    }
}

template <int NLD>
__device__ __forceinline__ void load_tile_fast(float4 (&r)[NLD], const float* origin, const int (&off)[NLD]) {
    const gptr4 b = opaque_base(origin);
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
        const v4f v = b[off[p]];
        r[p] = make_float4(v.x, v.y, v.z, v.w);
    }
}

template <bool ROWK, int BK, int XT>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const RegTile<BK, XT>& r, int tid) {
    constexpr int KQ = BK / 4, RPP = 256 / KQ, XQ = XT / 4, KPP = 256 / XQ;
#pragma unroll
    for (int p = 0; p < Tile<BK, XT>::NLD; ++p) {
        if (ROWK)
            *reinterpret_cast<float4*>(s + (p * RPP + tid / KQ) * Tile<BK, XT>::ROW_STRIDE + 4 * (tid % KQ)) = r[p];
        else
            *reinterpret_cast<float4*>(s + (p * KPP + tid / XQ) * Tile<BK, XT>::COL_STRIDE + 4 * (tid % XQ)) = r[p];
    }
}

// LDS -> registers: the A (MI row tiles) and B (2 column tiles) fragments of one 8-wide k group.
template <bool TA, bool TB, int BK, int MI>
__device__ __forceinline__ void read_frags(float (&af)[MI][4], float (&bf)[2][4], const float* __restrict__ a_s,
                                           const float* __restrict__ b_s, int q, int wm, int wn, int li, int hi) {
    constexpr int BMT = 64 * MI;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wm * (32 * MI) + i * 32 + li;
        if (!TA) {
            const float4 v = *reinterpret_cast<const float4*>(a_s + row * Tile<BK, BMT>::ROW_STRIDE + 8 * q + 4 * hi);
            af[i][0] = v.x, af[i][1] = v.y, af[i][2] = v.z, af[i][3] = v.w;
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) af[i][s] = a_s[(8 * q + 4 * hi + s) * Tile<BK, BMT>::COL_STRIDE + row];
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + li;
        if (TB) {
            const float4 v = *reinterpret_cast<const float4*>(b_s + col * Tile<BK, BN>::ROW_STRIDE + 8 * q + 4 * hi);
            bf[j][0] = v.x, bf[j][1] = v.y, bf[j][2] = v.z, bf[j][3] = v.w;
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) bf[j][s] = b_s[(8 * q + 4 * hi + s) * Tile<BK, BN>::COL_STRIDE + col];
        }
    }
}

// TA: A is stored [K][M] (op(A) = A^T).  TB: B is stored [N][K] (op(B) = B^T).
// MI = 32-row MFMA tiles per wave along M: block tile (64 MI) x 128, wave tile (32 MI) x 64.
template <bool TA, bool TB, int BK, int MI>
__global__ __launch_bounds__(256, (BK == 16 && MI == 2) ? 4 : 2) void gemm_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 64 * MI;
    constexpr int A_TILE = TA ? Tile<BK, BM>::COL_TILE : Tile<BK, BM>::ROW_TILE;
    constexpr int B_TILE = TB ? Tile<BK, BN>::ROW_TILE : Tile<BK, BN>::COL_TILE;
    float* sA = reinterpret_cast<float*>(smem);
    float* sB = sA + 2 * A_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, hi = lane >> 5;

    int bm, bn;
    tile_of_block(g, blockIdx.x, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;
    const int kbeg = blockIdx.z * g.k_split_len;
    const int kend = min(g.K, kbeg + g.k_split_len);
    const int nk = (kend - kbeg + BK - 1) / BK;
    float* __restrict__ C = g.C + (int64_t)blockIdx.z * g.c_split_stride;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Software pipeline (registers hold the tile that is ONE ahead of the LDS stage being computed):
    //   iteration kt:  MFMAs(group 0 of stage cur) -> ds_write(regs = tile kt+1 -> stage cur^1)
    //                  -> global loads(tile kt+2 -> regs) -> MFMAs(groups 1..) -> barrier
    // so the global loads get a full k-tile of latency budget, the LDS writes sit in the shadow of the
    // MFMAs, and the end of a k-tile is a bare barrier (no vmcnt wait, no LDS write) -- the exposed
    // "wait, write, barrier" tail of the classic double-buffer loop cost ~15 % here.
    RegTile<BK, BM> ra;
    RegTile<BK, BN> rb;
    OffTile<BK, BM> offa;
    OffTile<BK, BN> offb;
    tile_offsets<!TA, BK, BM>(offa, g.lda, m0, g.M, tid);
    tile_offsets<TB, BK, BN>(offb, g.ldb, n0, g.N, tid);
    // tile origins: "row" tiles advance by BK columns per k-tile, "col" tiles by BK rows
    const float* a_org = TA ? g.A + (int64_t)kbeg * g.lda + m0 : g.A + (int64_t)m0 * g.lda + kbeg;
    const float* b_org = TB ? g.B + (int64_t)n0 * g.ldb + kbeg : g.B + (int64_t)kbeg * g.ldb + n0;
    const int64_t a_step = TA ? (int64_t)BK * g.lda : BK, b_step = TB ? BK : (int64_t)BK * g.ldb;
    load_tile<!TA, BK, BM>(ra, g.A, g.lda, m0, kbeg, g.M, kend, tid);
    load_tile<TB, BK, BN>(rb, g.B, g.ldb, n0, kbeg, g.N, kend, tid);
    store_tile<!TA, BK, BM>(sA, ra, tid);
    store_tile<TB, BK, BN>(sB, rb, tid);
    if (nk > 1) {
        load_tile<!TA, BK, BM>(ra, g.A, g.lda, m0, kbeg + BK, g.M, kend, tid);
        load_tile<TB, BK, BN>(rb, g.B, g.ldb, n0, kbeg + BK, g.N, kend, tid);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const float* a_s = sA + cur * A_TILE;
        const float* b_s = sB + cur * B_TILE;
        // fragment reads run one 8-wide k group AHEAD of the MFMAs that consume them
        float af[2][MI][4], bf[2][2][4];
        read_frags<TA, TB, BK, MI>(af[0], bf[0], a_s, b_s, 0, wm, wn, li, hi);
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const int cq = q & 1;
            if (q + 1 < BK / 8) read_frags<TA, TB, BK, MI>(af[cq ^ 1], bf[cq ^ 1], a_s, b_s, q + 1, wm, wn, li, hi);
            __builtin_amdgcn_sched_barrier(0);  // keep the look-ahead reads ABOVE this group's MFMAs
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cq][i][s], bf[cq][j][s], acc[i][j], 0, 0, 0);
            if (q == 0) {
                if (kt + 1 < nk) {
                    store_tile<!TA, BK, BM>(sA + (cur ^ 1) * A_TILE, ra, tid);
                    store_tile<TB, BK, BN>(sB + (cur ^ 1) * B_TILE, rb, tid);
                }
                if (kt + 2 < nk) {
                    const int k2 = kbeg + (kt + 2) * BK;
                    if (k2 + BK <= kend) {      // full k-tile: unguarded fast path
                        load_tile_fast(ra, a_org + (kt + 2) * a_step, offa);
                        load_tile_fast(rb, b_org + (kt + 2) * b_step, offb);
                    } else {                    // ragged last k-tile: guarded, zero-filled
                        load_tile<!TA, BK, BM>(ra, g.A, g.lda, m0, k2, g.M, kend, tid);
                        load_tile<TB, BK, BN>(rb, g.B, g.ldb, n0, k2, g.N, kend, tid);
                    }
                }
            }
        }
        AVSI_LDS_BARRIER();  // LDS visibility only: the tile-(kt+2) global loads stay in flight across it
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const bool accumulate = g.beta != 0.f;  // uniform: keeps the read-modify-write out of the common path
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + li;
        bv[j] = (g.bias && col < g.N) ? g.bias[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row >= g.M) continue;
            int64_t orow = row;
            if (g.row_map_bp > 0) {  // time-major (t, b) -> batch-major (b, t), padded batch rows dropped
                const int t = row / g.row_map_bp, b = row - t * g.row_map_bp;
                if (b >= g.row_map_b) continue;
                orow = (int64_t)b * g.row_map_t + t;
            }
            const float rsc = g.row_scale ? g.row_scale[row] : 1.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + li;
                if (col >= g.N) continue;
                float* c = C + orow * g.ldc + col;
                float v = (g.alpha * acc[i][j][r] + bv[j]) * rsc;
                if (accumulate) v += g.beta * *c;
                *c = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// LDS-DMA variant (128 x 128 x 16 tiles, 3-stage ring).  Operand tiles go global -> LDS with
// global_load_lds_dwordx4: no VGPR staging, no ds_write, nothing for the waves to wait on but a
// counted s_waitcnt vmcnt(4) that leaves the NEXT tile's four DMAs in flight across the barrier.
// Measured against the register-staged loop (M = 524288, N = 2048, TFLOP/s): K=256 115 vs 110,
// K=512 131 vs 126, K=4096 135 vs 132; A^T B (weight gradients) 32.5 vs 28.5 unsplit.  A B^T is
// a tie (127.6 vs 128.9) and stays on the register path.  What still separates this from the
// 148 TFLOP/s the same loop reaches with the DMA issue removed is not latency (dropping the
// vmcnt wait changes nothing), not HBM traffic (an L2-resident source changes nothing) and not
// operand data (all-zero inputs change nothing); tools/mfma_f32_lds.hip modes 7/8 reproduce it.
// The DMA writes LDS linearly (wave-uniform base + lane * 16 B), so the LDS images are unpadded:
//   "col" tile [16][128]: one instruction = two k-rows; fragment reads are ds_read_b32, lanes along x;
//   "row" tile [128][16]: one instruction = 16 rows of 64 B; bank conflicts of the ds_read_b128
//     fragment reads are removed by an XOR swizzle of the 16-byte chunk index with (row >> 2) & 3,
//     applied on the SOURCE address of each lane (same 64-byte segment, coalescing unchanged) and
//     on the read address -- never on the LDS destination.
// Needs the reduction length to be a multiple of 16; other shapes take the register-staged kernel.
// ------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gvoid_t;
typedef __attribute__((address_space(3))) void* lvoid_t;

// One 16-byte-per-lane global -> LDS DMA (LDS address = M0 + lane * 16).  Written as inline asm
// rather than __builtin_amdgcn_global_load_lds on purpose: for the builtin the compiler cannot
// prove that later ds_reads do not alias the DMA's destination and puts s_waitcnt vmcnt(0) in
// front of them, which serialises every tile on the DMA just issued.  The kernel counts vmcnt
// itself (the DMAs are the only vector-memory operations in its main loop).  M0 is a reserved
// register the compiler does not allocate; nothing else in this kernel uses it.
__device__ __forceinline__ void dma16(const float* src, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory");
}

// 4 x 4 transpose inside every quad of lanes: on return register c of lane b (b = lane & 3) holds what register b of lane c held.
// Two exchanges: with lane ^ 1 over the register pairs (0, 1) and (2, 3), then with lane ^ 2 over (0, 2) and (1, 3).
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void quad_transpose(float& r0, float& r1, float& r2, float& r3, bool odd1, bool odd2) {
    float s = odd1 ? r0 : r1, t = odd1 ? r2 : r3;
    s = quad_perm<0xB1>(s), t = quad_perm<0xB1>(t);         // quad_perm [1, 0, 3, 2]
    r0 = odd1 ? s : r0, r1 = odd1 ? r1 : s, r2 = odd1 ? t : r2, r3 = odd1 ? r3 : t;
    float u = odd2 ? r0 : r2, v = odd2 ? r1 : r3;
    u = quad_perm<0x4E>(u), v = quad_perm<0x4E>(v);         // quad_perm [2, 3, 0, 1]
    r0 = odd2 ? u : r0, r1 = odd2 ? v : r1, r2 = odd2 ? r2 : u, r3 = odd2 ? r3 : v;
}

template <bool TA, bool TB, int BK, int NST, bool CONV = false, int BNT = 128, bool TAIL = false, bool PERS = false>
__global__ __launch_bounds__(256, BNT == 256 ? 2 : (BK == 16 ? (NST == 3 ? 3 : 4) : 2)) void gemm_dma_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!TAIL || (BNT == 256 && !TA && !TB && !CONV && BK == 16), "the folded last column exists for A . B on the 256-wide tile");
    // PERS (round 6): the grid is the RESIDENT workgroups (two per CU) and a workgroup takes tile after tile (GemmArgs::tile_ctr).  What that
    // buys is the seam between two tiles: the next tile's first two k-tiles are requested BEFORE this tile's 128 stores per wave (512 per workgroup)
    // are issued, so the DMA latency, and the end of a workgroup / start of the next one that used to sit there (store
    // acknowledgement, teardown, launch, offsets, prologue: ~7 us of a 58 us tile at K = 272), run under the stores.  Measured
    // with the stores taken out (AVSI_GEMM_DIAG=1): 17.5 -> 15.4 ms at K = 272, 31.5 -> 29.9 at 512 -- the stores cost 1.6 - 2.1 ms a
    // launch whatever K, i.e. about what 16.8 GB take to write at the memory's rate, nothing of it hidden.
    static_assert(!PERS || (BNT == 256 && !TA && !TB && !CONV && BK == 16 && NST == 3), "persistent form: A . B on the 256-wide tile");
    static_assert((BK == 16 && (NST == 3 || NST == 2)) || (BK == 32 && NST == 2), "tile shapes this kernel was tuned for");
    static_assert(!CONV || (!TB && BK == 16), "the implicit-GEMM gathers are written for A . B and A^T . B with 16-deep tiles");
    static_assert(BNT == 128 || ((BNT == 64 || BNT == 32 || BNT == 256) && !TB && BK == 16 && NST == 3),
                  "narrow / wide N tiles exist for the 16-deep forms with B as [k][n]");
    // wave grid WM x WN, each wave TM x TN MFMA tiles: 128 x 128 = (2 x 2) x (2 x 2), 128 x 64 = (2 x 2) x (2 x 1),
    // 128 x 32 = (4 x 1) x (1 x 1)
    // 128 x 256 = (2 x 2) x (2 x 4): half the barriers and a quarter less LDS traffic per MFMA than 128 x 128
    constexpr int WN = BNT == 32 ? 1 : 2, TN = BNT == 256 ? 4 : (BNT == 128 ? 2 : 1), TM = BNT == 32 ? 1 : 2;
    constexpr int BM = 128, TILE = 128 * BK, TILEB = BNT * BK;
    constexpr int PPWB = BNT == 256 ? 4 : (BNT == 128 ? BK / 8 : 1);   // B pieces per wave (the 2 pieces of a 32-wide tile are issued twice)
    constexpr int PPW = BK / 8;         // 1-KiB DMA pieces per wave and operand tile
    constexpr int CPR = BK / 4;         // 16-byte chunks per row of a "row" tile
    constexpr int RPP = 256 / BK;       // rows of a "row" tile per DMA piece
    // XOR swizzle of the chunk index inside a row: spreads the ds_read_b128 of 8 neighbouring rows over all banks
    auto swz = [](int row) { return BK == 16 ? (row >> 2) & 3 : row & 7; };
    float* sA = reinterpret_cast<float*>(smem);
    float* sB = sA + NST * TILE;   // NST stages of TILEB floats
    float* sW = sB + NST * TILEB;  // TAIL: column g.tail_col of B, K floats
    const uint32_t lds_a = (uint32_t)(uintptr_t)(lvoid_t)smem, lds_b = lds_a + NST * TILE * 4u;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;

    // PERS: tiles are taken from the counter of this workgroup's XCD class, two ahead: the index of the tile after next is requested
    // before the current tile's stores and read behind the next tile's first wait, so its round trip costs nothing.  (Also built: the
    // claim four k-tiles before the end of the current tile, i.e. late -- the tiles in flight then stay as consecutive as under
    // hardware dispatch and the launch fetches 16.9 GB at K = 512 instead of 23.2 (one workgroup per tile: 16.1; a fixed stride: 26.6),
    // but the branch in the k loop cost 20 - 52 bytes of scratch and most of the gain: 17.53 / 31.08 ms against 17.23 / 30.93 for this
    // form and 17.91 / 31.44 for one workgroup per tile.  The kernel is bound by the matrix pipes at 1.3 TB/s of traffic: speed won.)
    __shared__ int s_take[2];
    const int xcls = blockIdx.x % AVSI_NUM_XCD;
    const int class_tiles = (g.m_blocks * g.n_blocks) / AVSI_NUM_XCD + (xcls < (g.m_blocks * g.n_blocks) % AVSI_NUM_XCD ? 1 : 0);
    int idx_next = 0, idx_fetched = 0;
    int tile_id = blockIdx.x;
    if (PERS) {
        if (tid == 0) {
            s_take[0] = (int)atomicAdd(g.tile_ctr + xcls, 1u);
            s_take[1] = (int)atomicAdd(g.tile_ctr + xcls, 1u);
        }
        __syncthreads();
        const int idx_cur = s_take[0];
        idx_next = s_take[1];
        if (idx_cur >= class_tiles) return;              // (more resident workgroups than tiles of this class)
        tile_id = xcls + AVSI_NUM_XCD * idx_cur;
    }
    int bm, bn;
    tile_of_block(g, tile_id, bm, bn);
    int m0 = bm * BM, n0 = bn * BNT;
    const int kbeg = blockIdx.z * g.k_split_len;
    const int kend = min(g.K, kbeg + g.k_split_len);
    const int nk = (kend - kbeg) / BK;
    float* __restrict__ C = g.C + (int64_t)blockIdx.z * g.c_split_stride;

    // per-lane source offsets (floats, relative to the tile origin) of this wave's DMA pieces per operand
    constexpr int PPM = PPW > PPWB ? PPW : PPWB;
    int64_t offa[PPM], offb[PPM];
    const int pb0 = BNT >= 128 ? wave * PPWB : (BNT == 64 ? wave : (wave & 1));   // first B piece of this wave
    auto place = [&]() {          // (for the tile at m0, n0)
#pragma unroll
        for (int j = 0; j < PPM; ++j) {
            const int p = wave * PPW + j;  // 1-KiB piece of the tile
            if (j >= PPW) {
            } else if (!TA) {  // row tile [m][k]
                const int row = p * RPP + lane / CPR, cl = (lane % CPR) ^ swz(row);
                offa[j] = (int64_t)(min(m0 + row, g.M - 1) - m0) * g.lda + cl * 4;
            } else {    // col tile [k][m]
                const int x = min(m0 + 4 * (lane & 31), ((g.M + 3) & ~3) - 4) - m0;
                offa[j] = (int64_t)(2 * p + (lane >> 5)) * g.lda + x;
            }
            if (TB) {   // row tile [n][k]
                const int row = p * RPP + lane / CPR, cl = (lane % CPR) ^ swz(row);
                offb[j] = (int64_t)(min(n0 + row, g.N - 1) - n0) * g.ldb + cl * 4;
            } else if (j < PPWB) {    // col tile [k][n]: one piece = 256 / BNT k-rows of BNT floats
                constexpr int XQ = BNT / 4;
                const int x = min(n0 + 4 * (lane % XQ), ((g.N + 3) & ~3) - 4) - n0;
                offb[j] = (int64_t)((pb0 + j) * (256 / BNT) + lane / XQ) * g.ldb + x;
            }
        }
    };
    place();
    // implicit GEMM: output pixel (b, h, w) of this lane's rows, and its 16-byte chunk inside a 16-deep k tile
    int cvh[PPW], cvw[PPW], cvb[PPW], cvcl[PPW];
    bool cvok[PPW];
    if (CONV && !TA) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int row = (wave * PPW + j) * RPP + lane / CPR;
            const int m = m0 + row;
            cvok[j] = m < g.M;
            const int mm = cvok[j] ? m : 0;
            cvw[j] = mm % g.cW;
            cvh[j] = (mm / g.cW) % g.cH;
            cvb[j] = mm / (g.cW * g.cH);
            cvcl[j] = ((lane % CPR) ^ swz(row)) * 4;
        }
    }
    // filter-gradient form (A^T . B, A = the im2col matrix, reduction over pixels): this lane's 4 columns
    // m = (tap, c .. c+3) are fixed for the whole kernel, the pixel changes with the k row
    int wt_dh = 0, wt_dw = 0, wt_c = 0;
    bool wt_ok = false, wt_from0 = true;
    if (CONV && TA) {
        const int Ct = g.cC0 + g.cC1, m = m0 + 4 * (lane & 31);
        wt_ok = m < g.M;
        const int tap = wt_ok ? m / Ct : 0;
        wt_c = wt_ok ? m - tap * Ct : 0;
        wt_dh = tap / g.ck - g.ck / 2, wt_dw = tap % g.ck - g.ck / 2;
        wt_from0 = wt_c < g.cC0;
    }
    const float* a_org = TA ? g.A + (int64_t)kbeg * g.lda + m0 : g.A + (int64_t)m0 * g.lda + kbeg;
    const float* b_org = TB ? g.B + (int64_t)n0 * g.ldb + kbeg : g.B + (int64_t)kbeg * g.ldb + n0;
    const int64_t a_step = TA ? (int64_t)BK * g.lda : BK, b_step = TB ? BK : (int64_t)BK * g.ldb;

    auto issue = [&](int kt) {
        const int st = kt % NST;
        const float* ao = a_org + kt * a_step;
        const float* bo = b_org + kt * b_step;
        // implicit GEMM: the whole k tile lies in one filter tap and one source (channel counts % 16 == 0)
        int cdh = 0, cdw = 0, cc = 0;
        bool from0 = true;
        if (CONV && !TA) {
            const int Ct = g.cC0 + g.cC1, k0 = kbeg + kt * BK;
            const int tap = k0 / Ct;
            cc = k0 - tap * Ct;
            cdh = tap / g.ck - g.ck / 2, cdw = tap % g.ck - g.ck / 2;
            from0 = cc < g.cC0;
        }
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const float* asrc = ao + offa[j];
            if (CONV && TA) {
                const int px = kbeg + kt * BK + 2 * (wave * PPW + j) + (lane >> 5);     // pixel of this lane's k row
                const int w = px % g.cW, h = (px / g.cW) % g.cH, b = px / (g.cW * g.cH);
                const int hh = h + wt_dh, ww = w + wt_dw;
                const bool ok = wt_ok && px < kend && hh >= 0 && hh < g.cH && ww >= 0 && ww < g.cW;
                if (!ok)
                    asrc = g.conv_zeros;
                else if (wt_from0)
                    asrc = g.conv_s0 + (((int64_t)b * g.cH + hh) * g.cW + ww) * g.cld0 + wt_c;
                else
                    asrc = g.conv_s1 + (((int64_t)b * (g.cH >> 1) + (hh >> 1)) * (g.cW >> 1) + (ww >> 1)) * g.cld1 +
                           (wt_c - g.cC0);
            }
            if (CONV && !TA) {
                const int hh = cvh[j] + cdh, ww = cvw[j] + cdw;
                const bool ok = cvok[j] && hh >= 0 && hh < g.cH && ww >= 0 && ww < g.cW;
                if (!ok)
                    asrc = g.conv_zeros + cvcl[j];
                else if (from0)
                    asrc = g.conv_s0 + (((int64_t)cvb[j] * g.cH + hh) * g.cW + ww) * g.cld0 + cc + cvcl[j];
                else
                    asrc = g.conv_s1 + (((int64_t)cvb[j] * (g.cH >> 1) + (hh >> 1)) * (g.cW >> 1) + (ww >> 1)) * g.cld1 +
                           (cc - g.cC0) + cvcl[j];
            }
            dma16(asrc, lds_a + (uint32_t)(st * TILE + (wave * PPW + j) * 256) * 4u);
            if (j < PPWB) dma16(bo + offb[j], lds_b + (uint32_t)(st * TILEB + (pb0 + j) * 256) * 4u);
        }
#pragma unroll
        for (int j = PPW; j < PPWB; ++j) dma16(bo + offb[j], lds_b + (uint32_t)(st * TILEB + (pb0 + j) * 256) * 4u);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments of one 8-deep k group (q < BK / 8) of a staged tile pair
    auto rd = [&](int q, const float* a_s, const float* b_s, float (&a)[TM][4], float (&b)[TN][4]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * (32 * TM) + i * 32 + li;
            if (!TA) {
                const float4 v = *reinterpret_cast<const float4*>(a_s + row * BK + (((2 * q + hi) ^ swz(row)) << 2));
                a[i][0] = v.x, a[i][1] = v.y, a[i][2] = v.z, a[i][3] = v.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) a[i][s] = a_s[(8 * q + 4 * hi + s) * 128 + row];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * (32 * TN) + j * 32 + li;
            if (TB) {
                const float4 v = *reinterpret_cast<const float4*>(b_s + col * BK + (((2 * q + hi) ^ swz(col)) << 2));
                b[j][0] = v.x, b[j][1] = v.y, b[j][2] = v.z, b[j][3] = v.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) b[j][s] = b_s[(8 * q + 4 * hi + s) * BNT + col];
            }
        }
    };
    auto mm = [&](const float (&a)[TM][4], const float (&b)[TN][4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    };

    // TAIL: the folded column of B into LDS (a strided gather of K floats per workgroup, L2 hits after the first workgroups),
    // requested in front of the first tiles' DMAs so that it has landed when they have
    float wcol[4] = {0.f, 0.f, 0.f, 0.f};
    if (TAIL) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (tid + 256 * j < g.K) wcol[j] = g.B[(int64_t)(tid + 256 * j) * g.ldb + g.tail_col];
    }
    float tail_acc = 0.f;
    // one 16-deep k tile of the folded column: thread (row = tid / 2, half = tid % 2) multiplies its 8 staged A values
    auto tail_dot = [&](int kt_, const float* a_s) {
        const int row = tid >> 1, h = tid & 1;
        const float4 x0 = *reinterpret_cast<const float4*>(a_s + row * BK + (((2 * h) ^ swz(row)) << 2));
        const float4 x1 = *reinterpret_cast<const float4*>(a_s + row * BK + (((2 * h + 1) ^ swz(row)) << 2));
        const float4 w0 = *reinterpret_cast<const float4*>(sW + kt_ * BK + 8 * h);
        const float4 w1 = *reinterpret_cast<const float4*>(sW + kt_ * BK + 8 * h + 4);
        tail_acc = fmaf(x0.x, w0.x, tail_acc), tail_acc = fmaf(x0.y, w0.y, tail_acc);
        tail_acc = fmaf(x0.z, w0.z, tail_acc), tail_acc = fmaf(x0.w, w0.w, tail_acc);
        tail_acc = fmaf(x1.x, w1.x, tail_acc), tail_acc = fmaf(x1.y, w1.y, tail_acc);
        tail_acc = fmaf(x1.z, w1.z, tail_acc), tail_acc = fmaf(x1.w, w1.w, tail_acc);
    };

    // ring of NST stages: tiles kt+1 .. kt+NST-1 are in flight / landed while tile kt is computed
    constexpr int PPT = PPW + PPWB;     // DMA instructions per wave and k-tile
    if (nk > 0) issue(0);
    if (NST == 3 && nk > 1) issue(1);
    bool first_tile = true;
    float af[2][TM][4], bf[2][TN][4];
#pragma unroll 1
    for (;;) {      // (one pass unless PERS)
    if (NST == 3 && nk > 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");      // (PERS, from the second tile on: this tile's first k-tile
    else                                                                //  AND all but the last few stores of the tile before)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (TAIL && first_tile) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (tid + 256 * j < g.K) sW[tid + 256 * j] = wcol[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (PERS && !first_tile) {          // the index requested in front of the last tile's stores has arrived (the wait above)
        if (tid == 0) s_take[0] = idx_fetched;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const bool had_fetch = PERS && !first_tile;
    first_tile = false;
    __builtin_amdgcn_s_barrier();
    if (had_fetch) idx_next = s_take[0];        // (read by everyone before thread 0 can write it again: a whole tile later)

    // tile kt+1 must have landed before anyone reads it; with three stages tile kt+2's DMAs (just issued) stay in flight
    // across the barrier
    auto tile_done = [&](int kt) {
        if (NST == 3 && kt + 2 < nk)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPT) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // The k-tiles in order, in up to three runs of whole tiles with a tile that ENDS in zero padding (g.sp_kt, ascending;
    // -1 = none) after the first two: such a tile runs a LOOP over its first sp_steps MFMA steps (fragments of one step
    // read with 4-byte LDS loads).  Loops only -- any branch that selects between two straight-line versions of a tile
    // made the compiler copy the 128 accumulator registers of the wide tile at the merge (256 VGPRs + 500 spilled).
    int kt = 0;
#pragma unroll 1
    for (int seg = 0; seg < 3; ++seg) {
        const int sp = (BK == 16 && !TA && !TB && !CONV && seg < 2) ? g.sp_kt[seg] : -1;
        const int run_end = sp >= 0 && sp < nk ? sp : nk;
        for (; kt < run_end; ++kt) {
            // the stage being refilled was last read (tile kt-1) before the barrier of iteration kt-1
            if (kt + NST - 1 < nk) issue(kt + NST - 1);
            const float* a_s = sA + (kt % NST) * TILE;
            const float* b_s = sB + (kt % NST) * TILEB;
#pragma unroll
            for (int h = 0; h < BK / 16; ++h) {
                rd(2 * h, a_s, b_s, af[0], bf[0]);
                rd(2 * h + 1, a_s, b_s, af[1], bf[1]);
                mm(af[0], bf[0]);
                if (TAIL) tail_dot(kt, a_s);        // (VALU work in the shadow of the MFMAs just issued)
                mm(af[1], bf[1]);
            }
            tile_done(kt);
        }
        if (kt >= nk) break;
        if (BK == 16 && !TA && !TB && !CONV) {
            if (kt + NST - 1 < nk) issue(kt + NST - 1);
            const float* a_s = sA + (kt % NST) * TILE;
            const float* b_s = sB + (kt % NST) * TILEB;
            const int steps = g.sp_steps[seg];
#pragma unroll 1
            for (int st = 0; st < steps; ++st) {          // step 4 q + s multiplies k = 8 q + s (lanes 0 .. 31) and 8 q + 4 + s
                const int q = st >> 2, s4 = st & 3;
                float a1[TM], b1[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * (32 * TM) + i * 32 + li;
                    a1[i] = a_s[row * BK + (((2 * q + hi) ^ swz(row)) << 2) + s4];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) b1[j] = b_s[(8 * q + 4 * hi + s4) * BNT + wn * (32 * TN) + j * 32 + li];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
            }
            if (TAIL) tail_dot(kt, a_s);            // (the rows of B behind the promise are zero in this column too: the whole tile, no step count)
            tile_done(kt);
            ++kt;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // PERS: the tile after this one -- its offsets, and the DMAs of its first two k-tiles (every stage of the ring is free: the
    // last barrier above is behind the last fragment read) -- BEFORE this tile's stores; m0 / n0 of the epilogue are kept aside
    const int em0 = m0, en0 = n0;
    bool more = false;
    if (PERS) {
        more = idx_next < class_tiles;
        if (more) {
            tile_id = xcls + AVSI_NUM_XCD * idx_next;
            if (tid == 0) idx_fetched = (int)atomicAdd(g.tile_ctr + xcls, 1u);
            tile_of_block(g, tile_id, bm, bn);
            m0 = bm * BM, n0 = bn * BNT;
            place();
            a_org = g.A + (int64_t)m0 * g.lda + kbeg;
            b_org = g.B + (int64_t)kbeg * g.ldb + n0;
            if (nk > 0) issue(0);
            if (nk > 1) issue(1);
        }
    }

    const bool accumulate = g.beta != 0.f;
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = en0 + wn * (32 * TN) + j * 32 + li;
        bv[j] = (g.bias && col < g.N && blockIdx.z == 0) ? g.bias[col] : 0.f;      // (split launches: the bias goes into slab 0)
    }
    if (TAIL) {
        // the folded column: the two halves of a row's dot product sit in neighbouring lanes
        const float dot = tail_acc + __shfl_xor(tail_acc, 1, 64);
        const int row = em0 + (tid >> 1);
        if (!(tid & 1) && row < g.M) {
            int64_t orow = row;
            bool live = true;
            if (g.row_map_bp > 0) {
                const int t = row / g.row_map_bp, b = row - t * g.row_map_bp;
                live = b < g.row_map_b;
                orow = (int64_t)b * g.row_map_t + t;
            }
            if (live) {
                const float rsc = g.row_scale ? g.row_scale[row] : 1.f;
                C[orow * g.ldc + g.tail_col] = (g.alpha * dot + (g.bias ? g.bias[g.tail_col] : 0.f)) * rsc;
            }
        }
    }
    // Fast path of the 128 x 256 tile (whole tile inside the matrix, plain C = alpha A.B + bias): one 64-bit add per
    // store instead of the general addressing / masking / row-map code below.  Measured on the three layer GEMMs
    // of the benchmark step (ms): 128-wide general 84.6 / fast 85.4, 256-wide general 86.1 / fast 82.9 -- so the
    // narrow tiles keep the general form.
    bool fast_done = false;
    if (BNT == 256 && g.row_map_bp == 0 && !g.row_scale && !accumulate && em0 + BM <= g.M && en0 + BNT <= g.N) {
        if (g.diag & 1) {                    // (timing experiment: what the stores of a tile cost; one lane keeps the arithmetic alive)
            if (acc[0][0][0] == 12345.678f) C[0] = acc[1][3][15];
            fast_done = true;
        }
        if (!fast_done && (g.diag & 8) && !(g.ldc & 3) && !(reinterpret_cast<uintptr_t>(C) & 15)) {
            // AVSI_GEMM_DIAG=8 (A/B, results right): 16-byte stores.  The idea: a wave can have 63 vector-memory operations in
            // flight (vmcnt is six bits), so 128 dword stores per wave and tile (512 per workgroup) might be bound by acknowledgement round trips.  The
            // C / D layout has a lane's four consecutive registers on four consecutive ROWS of one column; a 4 x 4 transpose inside
            // every quad of lanes (two DPP exchanges per register pair) turns them into four consecutive COLUMNS of one row: 32
            // stores of 16 bytes per wave.  MEASURED SLOWER, same box: K = 272 17.86 against 17.35 ms, K = 512 31.72 against 31.12
            // (an instruction then touches eight rows' 128-byte segments instead of two): the dword form below stays.
            const bool odd1 = lane & 1, odd2 = lane & 2;
            float* cw = C + (int64_t)(em0 + wm * (32 * TM) + 4 * hi + (lane & 3)) * g.ldc + en0 + wn * (32 * TN) + 4 * (li >> 2);
            const int64_t ld8 = (int64_t)8 * g.ldc;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {                  // rows 8 rq + 4 hi + (0..3)
                    float* cr = cw + (int64_t)(i * 32) * g.ldc + rq * ld8;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float r0 = g.alpha * acc[i][j][4 * rq + 0] + bv[j], r1 = g.alpha * acc[i][j][4 * rq + 1] + bv[j];
                        float r2 = g.alpha * acc[i][j][4 * rq + 2] + bv[j], r3 = g.alpha * acc[i][j][4 * rq + 3] + bv[j];
                        quad_transpose(r0, r1, r2, r3, odd1, odd2);
                        *reinterpret_cast<float4*>(cr + j * 32) = make_float4(r0, r1, r2, r3);
                    }
                }
            }
            fast_done = true;
        }
        if (!fast_done) {
        float* cw = C + (int64_t)(em0 + wm * (32 * TM) + 4 * hi) * g.ldc + en0 + wn * (32 * TN) + li;
        const int64_t ld4 = (int64_t)4 * g.ldc;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float* ci = cw + (int64_t)(i * 32) * g.ldc;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {                  // rows 8 rq + (0..3)
                float* cr = ci + 2 * rq * ld4;
#pragma unroll
                for (int r3 = 0; r3 < 4; ++r3) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float v = g.alpha * acc[i][j][4 * rq + r3] + bv[j];
                        if (g.diag & 2) __builtin_nontemporal_store(v, cr + j * 32);     // (A/B: streaming stores)
                        else cr[j * 32] = v;
                    }
                    cr += g.ldc;
                }
            }
        }
        }
        fast_done = true;
    }
    if (!fast_done) {
    float st1[TN], st2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) st1[j] = st2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = em0 + wm * (32 * TM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row >= g.M) continue;
            int64_t orow = row;
            if (g.row_map_bp > 0) {
                const int t = row / g.row_map_bp, b = row - t * g.row_map_bp;
                if (b >= g.row_map_b) continue;
                orow = (int64_t)b * g.row_map_t + t;
            }
            const float rsc = g.row_scale ? g.row_scale[row] : 1.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = en0 + wn * (32 * TN) + j * 32 + li;
                if (col >= g.N) continue;
                float* c = C + orow * g.ldc + col;
                float v = (g.alpha * acc[i][j][r] + bv[j]) * rsc;
                if (accumulate) v += g.beta * *c;
                *c = v;
                if (CONV) st1[j] += v, st2[j] += v * v;
            }
        }
    }
    if (CONV && g.stats) {       // wave-uniform.  Rows in a fixed order per lane, the two row halves by one shuffle: deterministic
        constexpr int WMV = 4 / WN;          // wave rows of the block
        float* part = g.stats + ((int64_t)(bm * WMV + wm) * 2) * g.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float t1 = st1[j] + __shfl_xor(st1[j], 32, 64), t2 = st2[j] + __shfl_xor(st2[j], 32, 64);
            const int col = en0 + wn * (32 * TN) + j * 32 + li;
            if (hi == 0 && col < g.N) part[col] = t1, part[g.N + col] = t2;
        }
    }
    }
    if (!PERS || !more) break;
    // the next tile starts from zero
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    tail_acc = 0.f;
    }   // tiles of this workgroup
}

// Diagnostic switches, read once per process (a getenv per launch is a linear scan of the environment, and a small-batch
// step issues ten GEMMs): AVSI_GEMM_BK / _MI / _DMA / _BNT / _NGROUP, -1 = not set.
struct GemmEnv {
    int bk, mi, dma, bnt, ngroup, diag;
    static int read(const char* name) {
        const char* e = getenv(name);
        return e ? atoi(e) : -1;
    }
    GemmEnv() : bk(read("AVSI_GEMM_BK")), mi(read("AVSI_GEMM_MI")), dma(read("AVSI_GEMM_DMA")), bnt(read("AVSI_GEMM_BNT")),
                ngroup(read("AVSI_GEMM_NGROUP")), diag(read("AVSI_GEMM_DIAG")) {}
};
static const GemmEnv& gemm_env() {
    static const GemmEnv e;
    return e;
}

template <bool TA, bool TB>
int launch_dma(const GemmArgs& g, int splits, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * 3 * 128 * 16 * 4;   // 48 KiB: three stages of A and B
    if (gemm_env().dma == 2 && g.k_split_len % 32 == 0 && g.K % 32 == 0) {
        hipLaunchKernelGGL((gemm_dma_kernel<TA, TB, 32, 2>), dim3(g.m_blocks * g.n_blocks, 1, splits), dim3(256),
                           2 * 2 * 128 * 32 * 4, st, g);
        return avsi_launch_status();
    }
    if (gemm_env().dma == 3) {
        hipLaunchKernelGGL((gemm_dma_kernel<TA, TB, 16, 2>), dim3(g.m_blocks * g.n_blocks, 1, splits), dim3(256),
                           2 * 2 * 128 * 16 * 4, st, g);
        return avsi_launch_status();
    }
    if (!TA && !TB && g.bnt == 256) {
        // 72 KiB: two workgroups per CU (+ K floats for the folded last column).  AVSI_GEMM_PERSIST=0: one workgroup per tile
        // (A/B); the persistent form needs enough tiles per resident workgroup to pay for its fixed order
        // Not in a process that has been given a SHARE of the chip (AVSI_COOP_CUS < 256: several processes on one GPU, the
        // rehearsals of tests/): resident workgroups hold the LDS of every CU for the whole launch, and another process's
        // cooperative recurrent group, which needs whole CUs of one XCD at once, then waits for launch after launch
        // (tests/test_bench_contract_gpu.py with four ranks on one GPU ran into its 2 s bound)
        static const bool persist = !(getenv("AVSI_GEMM_PERSIST") && atoi(getenv("AVSI_GEMM_PERSIST")) == 0) &&
                                    !(getenv("AVSI_COOP_CUS") && atoi(getenv("AVSI_COOP_CUS")) > 0 && atoi(getenv("AVSI_COOP_CUS")) < AVSI_NUM_CU);
        const size_t lds_t = (size_t)3 * (128 + 256) * 16 * 4 + (g.tail_col >= 0 ? (size_t)g.K * 4 : 0);
        const int tiles = g.m_blocks * g.n_blocks, resident = 2 * AVSI_NUM_CU;
        const bool pers = persist && splits == 1 && tiles >= 4 * resident;
        GemmArgs gp = g;
        if (pers) {
            static unsigned* ctr_base[16] = {};
            static std::atomic<unsigned> slot_seq{0};
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return AVSI_ERR_LAUNCH;
            if (!ctr_base[dev] && hipGetSymbolAddress((void**)&ctr_base[dev], HIP_SYMBOL(avsi_gemm_tile_ctr)) != hipSuccess)
                return AVSI_ERR_LAUNCH;
            gp.tile_ctr = ctr_base[dev] + (size_t)(slot_seq.fetch_add(1) & 255u) * 16;
            if (hipMemsetAsync(gp.tile_ctr, 0, 64, st) != hipSuccess) return AVSI_ERR_LAUNCH;
        }
#define AVSI_WIDE_LAUNCH(TAILV, PERSV)                                                                                            \
    do {                                                                                                                          \
        (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<false, false, 16, 3, false, 256, TAILV, PERSV>,                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);                                        \
        hipLaunchKernelGGL((gemm_dma_kernel<false, false, 16, 3, false, 256, TAILV, PERSV>),                                      \
                           dim3((PERSV) ? resident : tiles, 1, (PERSV) ? 1 : splits), dim3(256), lds_t, st, gp);                  \
    } while (0)
        if (g.tail_col >= 0) {
            if (pers) AVSI_WIDE_LAUNCH(true, true);
            else AVSI_WIDE_LAUNCH(true, false);
        } else {
            if (pers) AVSI_WIDE_LAUNCH(false, true);
            else AVSI_WIDE_LAUNCH(false, false);
        }
#undef AVSI_WIDE_LAUNCH
        return avsi_launch_status();
    }
    if (!TB && g.bnt == 256) {
        constexpr size_t lds256 = (size_t)3 * (128 + 256) * 16 * 4;      // 72 KiB: two workgroups per CU
        (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<TA, false, 16, 3, false, 256>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
        hipLaunchKernelGGL((gemm_dma_kernel<TA, false, 16, 3, false, 256>), dim3(g.m_blocks * g.n_blocks, 1, splits), dim3(256),
                           lds256, st, g);
        return avsi_launch_status();
    }
    if (!TA && !TB && g.bnt == 32) {     // at most 32 columns (the last bin of the 257-bin projection): bound by reading A
        hipLaunchKernelGGL((gemm_dma_kernel<false, false, 16, 3, false, 32>), dim3(g.m_blocks * g.n_blocks, 1, splits), dim3(256),
                           (size_t)3 * (128 + 32) * 16 * 4, st, g);
        return avsi_launch_status();
    }
    if (!TA && !TB && g.bnt == 64) {
        hipLaunchKernelGGL((gemm_dma_kernel<false, false, 16, 3, false, 64>), dim3(g.m_blocks * g.n_blocks, 1, splits), dim3(256),
                           (size_t)3 * (128 + 64) * 16 * 4, st, g);
        return avsi_launch_status();
    }
    hipLaunchKernelGGL((gemm_dma_kernel<TA, TB, 16, 3>), dim3(g.m_blocks * g.n_blocks, 1, splits), dim3(256), lds, st, g);
    return avsi_launch_status();
}

template <bool TA, bool TB, int BK, int MI>
int launch(const GemmArgs& g, int splits, hipStream_t st) {
    constexpr int BM = 64 * MI;
    constexpr size_t lds = (size_t)2 * ((TA ? Tile<BK, BM>::COL_TILE : Tile<BK, BM>::ROW_TILE) +
                                        (TB ? Tile<BK, BN>::ROW_TILE : Tile<BK, BN>::COL_TILE)) * 4;
    static_assert(lds <= ((BK == 16 && MI == 2) ? 40 : 80) * 1024, "workgroups per CU must fit the 160 KiB LDS");
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)gemm_kernel<TA, TB, BK, MI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL((gemm_kernel<TA, TB, BK, MI>), dim3(g.m_blocks * g.n_blocks, 1, splits), dim3(256), lds, st, g);
    return avsi_launch_status();
}

template <int BK, int MI>
int dispatch(const GemmArgs& g, int transA, int transB, int splits, hipStream_t st) {
    if (!transA && !transB) return launch<false, false, BK, MI>(g, splits, st);
    if (!transA && transB) return launch<false, true, BK, MI>(g, splits, st);
    if (transA && !transB) return launch<true, false, BK, MI>(g, splits, st);
    return launch<true, true, BK, MI>(g, splits, st);
}

}  // namespace

// Internal entry shared with the other translation units (not part of the C ABI).
int avsi_gemm_launch(int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda,
                     const float* B, int64_t ldb, float beta, float* C, int64_t ldc, const avsi_gemm_epilogue* ep,
                     int splits, int64_t c_split_stride, hipStream_t st) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return AVSI_ERR_INVALID_ARG;
    // 16-byte global loads along each operand's contiguous dimension
    if ((lda & 3) || (ldb & 3) || (K & 3) || (reinterpret_cast<uintptr_t>(A) & 15) ||
        (reinterpret_cast<uintptr_t>(B) & 15))
        return AVSI_ERR_UNSUPPORTED;
    if (transA ? (lda < avsi_round_up(M, 4)) : (lda < K)) return AVSI_ERR_INVALID_ARG;
    if (transB ? (ldb < K) : (ldb < avsi_round_up(N, 4))) return AVSI_ERR_INVALID_ARG;
    if (ldc < N || splits < 1) return AVSI_ERR_INVALID_ARG;
    GemmArgs g{};
    g.A = A, g.B = B, g.C = C;
    g.bias = ep ? ep->bias : nullptr;
    g.row_scale = ep ? ep->row_scale : nullptr;
    g.row_map_bp = ep ? ep->row_map_bp : 0;
    g.row_map_t = ep ? ep->row_map_t : 0;
    g.row_map_b = ep ? ep->row_map_b : 0;
    if (g.row_map_bp < 0 || (g.row_map_bp > 0 && (g.row_map_t <= 0 || g.row_map_b <= 0 || M % g.row_map_bp)))
        return AVSI_ERR_INVALID_ARG;
    g.M = M, g.N = N, g.K = K;
    g.lda = lda, g.ldb = ldb, g.ldc = ldc;
    g.alpha = alpha, g.beta = beta;
    // tuning overrides (diagnostic): AVSI_GEMM_BK = 16 | 32, AVSI_GEMM_MI = 2 | 4
    const GemmEnv& env = gemm_env();
    const bool env_bk = env.bk >= 0, env_mi = env.mi >= 0;
    const int bk = env_bk ? env.bk : (transB ? 32 : 16);
    // A . B^T over many rows (the dX products of training): 256-row tiles, +2.7 % (129.7 -> 133.2 TFLOP/s)
    const int mi = env_mi ? env.mi : ((transB && !transA && M >= 65536) ? 4 : 2);
    const int BM = 64 * mi;
    g.m_blocks = (int)avsi_ceil_div(M, BM);
    // A . B whose last 128-wide tile would be at most half full and that is narrow enough for the tail to matter
    // (the 257-bin projection: 5 x 64 instead of 3 x 128 columns of MFMA work)
    g.k_split_len = (int)avsi_round_up(avsi_ceil_div(K, splits), 32);
    const bool dma_ok = !(transB && !transA) && (K % 16 == 0) && (g.k_split_len % 16 == 0) && !env_bk && !env_mi && env.dma != 0;
    g.bnt = (dma_ok && !transA && !transB && N < 1024 && ((N - 1) % BN) < 64 && !(env.dma > 1)) ? (N <= 32 ? 32 : 64) : BN;
    // wide layer GEMMs (N = 2048): 128 x 256 output tiles, each wave 64 x 128 -- half the barriers and a quarter less
    // LDS traffic per MFMA (K = 4096: 135 -> 142 TFLOP/s; the 512-deep layer GEMMs gain 2 %, the split-K weight
    // gradients 6 %), when that still leaves two workgroups per CU.  AVSI_GEMM_BNT=128: diagnostics
    static const int wide_min = getenv("AVSI_GEMM_WIDE_MIN") ? atoi(getenv("AVSI_GEMM_WIDE_MIN")) : 2 * AVSI_NUM_CU - 32;   // 8000 rows (32 utterances): 504 wide tiles, 111 -> 99 us
    // N = 256 exactly: the first 256 bins of the 257-bin projection, which the model issues on their own (round 3)
    if (dma_ok && !transB && g.bnt == BN && N % 256 == 0 && (N >= 1024 || N == 256) &&
        (int64_t)g.m_blocks * (N / 256) * splits >= wide_min && env.bnt != 128)
        g.bnt = 256;
    g.diag = env.diag > 0 ? env.diag : 0;
    g.tail_col = -1;
    // N = 257 over many rows (the projection onto the reference's 257 bins, models.py:119-122): the 256-wide tile for the first
    // 256 columns and the last one folded into the same workgroups (gemm_dma_kernel<.., TAIL>), instead of a second launch on
    // 32-wide tiles that reads all of A again for one column (0.87 ms of a 138 ms step at 8192 utterances).  AVSI_GEMM_FOLD_TAIL=0: off
    static const bool fold_tail = !(getenv("AVSI_GEMM_FOLD_TAIL") && atoi(getenv("AVSI_GEMM_FOLD_TAIL")) == 0);
    if (fold_tail && dma_ok && !transA && !transB && N == 257 && splits == 1 && beta == 0.f && K <= 1024 &&
        (int64_t)g.m_blocks >= wide_min && env.bnt != 128) {
        g.bnt = 256;
        g.tail_col = 256;
    }
    g.n_blocks = g.tail_col >= 0 ? 1 : (int)avsi_ceil_div(N, g.bnt);
    g.c_split_stride = splits > 1 ? c_split_stride : 0;
    g.sp_kt[0] = g.sp_kt[1] = -1, g.sp_steps[0] = g.sp_steps[1] = 8;
    if (ep && dma_ok && !transA && splits == 1) {
        // zero padding inside the reduction (the caller's promise): a k-tile that ends in it runs only the MFMA steps that
        // multiply something -- up to two such tiles (the 250 -> 256 padding of each half of a BLSTM layer's input)
        for (int k = 0; k < 4; ++k)
            if (ep->k_zero[k] < 0 || ep->k_zero[k] > K) return AVSI_ERR_INVALID_ARG;
        if (ep->k_zero[0] > ep->k_zero[1] || ep->k_zero[2] > ep->k_zero[3]) return AVSI_ERR_INVALID_ARG;
        int found = 0;
        for (int r = 0; r < 2 && found < 2; ++r) {
            const int lo = ep->k_zero[2 * r], hi = ep->k_zero[2 * r + 1];
            if (hi <= lo) continue;
            // the tile that holds lo, if the range reaches that tile's end
            const int kt = lo / 16, kv = lo - 16 * kt;
            if (hi < 16 * (kt + 1) && hi < K) continue;
            const int steps = kv == 0 ? 0 : (kv <= 4 ? kv : (kv <= 8 ? 4 : (kv <= 12 ? kv - 4 : 8)));
            if (steps < 8 && (found == 0 || g.sp_kt[0] != kt)) g.sp_kt[found] = kt, g.sp_steps[found] = steps, ++found;
        }
        if (found == 2 && g.sp_kt[0] > g.sp_kt[1]) {          // the kernel walks them in ascending order
            const int t0 = g.sp_kt[0], t1 = g.sp_steps[0];
            g.sp_kt[0] = g.sp_kt[1], g.sp_steps[0] = g.sp_steps[1], g.sp_kt[1] = t0, g.sp_steps[1] = t1;
        }
    }
    {   // column-group width: the group's slice of op(B), k_split_len x (n_group * 128) floats, should fill about half of
        // one XCD's 4 MiB L2

        const int64_t slice_bytes_per_block = (int64_t)g.k_split_len * g.bnt * 4;
        int ng = (int)((2 << 20) / (slice_bytes_per_block > 0 ? slice_bytes_per_block : 1));
        if (ng < 4) ng = 4;      // narrower groups re-read A more often than they save on B
        // (groups of equal width: 272 deep x 256 wide slices gave a width of 7 for 8 column blocks -- groups of 7 and 1, and layer 0
        //  took 17.4 ms where widths of 2 / 4 / 8 take 16.9 / 16.7 / 16.8: the next divisor of the block count below)
        if (ng < g.n_blocks)
            while (ng > 1 && g.n_blocks % ng) --ng;
        if (env.ngroup >= 0) ng = env.ngroup;
        g.n_group = ng < 1 ? 1 : (ng > g.n_blocks ? g.n_blocks : ng);
    }
    if ((int64_t)g.m_blocks * g.n_blocks > INT32_MAX) return AVSI_ERR_UNSUPPORTED;
    // "col" tiles read 4 consecutive x per lane and guard on the first: the ld padding up to a
    // multiple of 4 (checked above) keeps the tail addressable; such lanes feed unstored outputs.
    avsi_clear_error();
    if (dma_ok) {
        if (!transA && !transB) return launch_dma<false, false>(g, splits, st);
        if (!transA && transB) return launch_dma<false, true>(g, splits, st);
        if (transA && !transB) return launch_dma<true, false>(g, splits, st);
        return launch_dma<true, true>(g, splits, st);
    }
    if (mi == 4) return dispatch<16, 4>(g, transA, transB, splits, st);   // 256 x 128 tiles exist with BK = 16 only (LDS)
    return bk == 16 ? dispatch<16, 2>(g, transA, transB, splits, st) : dispatch<32, 2>(g, transA, transB, splits, st);
}

// Convolution as an implicit GEMM: out[(b,h,w)][n] = bias[n] + sum_{tap,c} in(b, h+dh, w+dw, c) filter[(tap, c)][n].
// number of [2][Cout] partial-statistics rows the launch below writes when `stats` is given
int avsi_conv2d_stats_parts(int B, int H, int W, int Cout) {
    const int64_t m_blocks = avsi_ceil_div((int64_t)B * H * W, 128);
    return (int)(m_blocks * (Cout <= 32 ? 4 : 2));
}

int avsi_conv2d_launch(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H, int W, int k,
                       const float* filter, int ldf, const float* bias, int Cout, float* out, int ldo, const float* zeros64,
                       float* stats, void* stream);

extern "C" int avsi_conv2d_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H,
                               int W, int k, const float* filter, int ldf, const float* bias, int Cout, float* out,
                               int ldo, const float* zeros64, void* stream) {
    return avsi_conv2d_launch(src0, C0, ld0, src1_coarse, C1, ld1, B, H, W, k, filter, ldf, bias, Cout, out, ldo, zeros64, nullptr,
                              stream);
}

int avsi_conv2d_launch(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H, int W, int k,
                       const float* filter, int ldf, const float* bias, int Cout, float* out, int ldo, const float* zeros64,
                       float* stats, void* stream) {
    if (!filter || !out || !zeros64 || B <= 0 || H <= 0 || W <= 0 || k < 1 || !(k & 1) || Cout <= 0 || C0 < 0 || C1 < 0)
        return AVSI_ERR_INVALID_ARG;
    if ((C0 && !src0) || (C1 && !src1_coarse) || ldf < Cout || ldo < Cout || (C0 && ld0 < C0) || (C1 && ld1 < C1))
        return AVSI_ERR_INVALID_ARG;
    // a 16-deep k tile must sit inside one tap of one source; 16-byte gathers
    if ((C0 & 15) || (C1 & 15) || C0 + C1 < 16 || (ld0 & 3) || (ld1 & 3) || (ldf & 3) || (C1 && ((H | W) & 1)))
        return AVSI_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(src0) | reinterpret_cast<uintptr_t>(src1_coarse) | reinterpret_cast<uintptr_t>(filter) |
         reinterpret_cast<uintptr_t>(zeros64)) & 15)
        return AVSI_ERR_UNSUPPORTED;
    const int64_t M64 = (int64_t)B * H * W;
    if (M64 > INT32_MAX) return AVSI_ERR_UNSUPPORTED;
    GemmArgs g{};
    g.A = zeros64, g.B = filter, g.C = out, g.bias = bias, g.row_scale = nullptr;
    g.M = (int)M64, g.N = Cout, g.K = k * k * (C0 + C1);
    g.lda = 0, g.ldb = ldf, g.ldc = ldo;
    g.alpha = 1.f, g.beta = 0.f;
    const int bnt = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);      // N tile: narrow layers do not pay for 128 columns
    g.m_blocks = (int)avsi_ceil_div(g.M, 128), g.n_blocks = (int)avsi_ceil_div(Cout, bnt);
    g.k_split_len = g.K, g.c_split_stride = 0, g.n_group = g.n_blocks;
    g.conv_s0 = src0, g.conv_s1 = src1_coarse, g.conv_zeros = zeros64;
    g.cH = H, g.cW = W, g.cC0 = C0, g.cld0 = ld0, g.cC1 = C1, g.cld1 = ld1, g.ck = k;
    g.stats = stats;
    if ((int64_t)g.m_blocks * g.n_blocks > INT32_MAX) return AVSI_ERR_UNSUPPORTED;
    avsi_clear_error();
    const dim3 grid(g.m_blocks * g.n_blocks), block(256);
    const hipStream_t st = (hipStream_t)stream;
    if (bnt == 32)
        hipLaunchKernelGGL((gemm_dma_kernel<false, false, 16, 3, true, 32>), grid, block, (size_t)3 * (128 + 32) * 16 * 4, st, g);
    else if (bnt == 64)
        hipLaunchKernelGGL((gemm_dma_kernel<false, false, 16, 3, true, 64>), grid, block, (size_t)3 * (128 + 64) * 16 * 4, st, g);
    else
        hipLaunchKernelGGL((gemm_dma_kernel<false, false, 16, 3, true, 128>), grid, block, (size_t)3 * (128 + 128) * 16 * 4, st, g);
    return avsi_launch_status();
}

int avsi_sum_slabs_launch(const float* slabs, int64_t n, int count, int64_t stride, float* out, float alpha,
                          hipStream_t st);

// The same convolution with the reduction over (tap, channel) cut into `splits` chunks (grid.z), partial slabs summed in
// chunk order by a second kernel (deterministic, like avsi_gemm_splitk_f32).  For the layers whose output is only a few
// tiles: at the reference's batch of 32 the 128-channel layers of the U-Net are 4 .. 64 output tiles on a 256-CU chip,
// each walking 36 .. 144 k-tiles on its own (134 us a launch whatever the size); cut to ~8 k-tiles per workgroup they
// fill the chip.  splits <= 1 or a workspace too small for the slabs: the plain launch.
extern "C" size_t avsi_conv2d_splitk_workspace_bytes(int B, int H, int W, int ldo, int splits) {
    if (B <= 0 || H <= 0 || W <= 0 || ldo <= 0 || splits < 2) return 0;
    return (size_t)splits * (size_t)B * H * W * (size_t)ldo * sizeof(float);
}

// splits that fill the chip about once with at least `min_ktiles` 16-deep k-tiles per workgroup (1 = do not split)
extern "C" int avsi_conv2d_splitk_suggest(int B, int H, int W, int k, int C0, int C1, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || k < 1 || Cout <= 0 || C0 + C1 <= 0) return 1;
    const int64_t M = (int64_t)B * H * W;
    const int bnt = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
    const int64_t tiles = avsi_ceil_div(M, 128) * avsi_ceil_div(Cout, bnt);
    const int ktiles = k * k * (C0 + C1) / 16;
    if (tiles >= AVSI_NUM_CU / 2 || ktiles < 16) return 1;
    int splits = (int)(AVSI_NUM_CU / tiles);
    const int max_by_depth = ktiles / 8;                 // at least 8 k-tiles per workgroup
    if (splits > max_by_depth) splits = max_by_depth;
    return splits < 2 ? 1 : splits;
}

extern "C" int avsi_conv2d_splitk_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H,
                                      int W, int k, const float* filter, int ldf, const float* bias, int Cout, float* out,
                                      int ldo, const float* zeros64, int splits, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    if (splits < 2 || ldo != Cout || !workspace || workspace_bytes < avsi_conv2d_splitk_workspace_bytes(B, H, W, ldo, splits))
        return avsi_conv2d_f32(src0, C0, ld0, src1_coarse, C1, ld1, B, H, W, k, filter, ldf, bias, Cout, out, ldo, zeros64, stream);
    if (!filter || !out || !zeros64 || B <= 0 || H <= 0 || W <= 0 || k < 1 || !(k & 1) || Cout <= 0 || C0 < 0 || C1 < 0)
        return AVSI_ERR_INVALID_ARG;
    if ((C0 && !src0) || (C1 && !src1_coarse) || ldf < Cout || ldo < Cout || (C0 && ld0 < C0) || (C1 && ld1 < C1))
        return AVSI_ERR_INVALID_ARG;
    if ((C0 & 15) || (C1 & 15) || C0 + C1 < 16 || (ld0 & 3) || (ld1 & 3) || (ldf & 3) || (C1 && ((H | W) & 1)))
        return AVSI_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(src0) | reinterpret_cast<uintptr_t>(src1_coarse) | reinterpret_cast<uintptr_t>(filter) |
         reinterpret_cast<uintptr_t>(zeros64) | reinterpret_cast<uintptr_t>(workspace)) & 15)
        return AVSI_ERR_UNSUPPORTED;
    const int64_t M64 = (int64_t)B * H * W;
    if (M64 > INT32_MAX) return AVSI_ERR_UNSUPPORTED;
    GemmArgs g{};
    g.A = zeros64, g.B = filter, g.C = static_cast<float*>(workspace), g.bias = bias, g.row_scale = nullptr;
    g.M = (int)M64, g.N = Cout, g.K = k * k * (C0 + C1);
    g.lda = 0, g.ldb = ldf, g.ldc = ldo;
    g.alpha = 1.f, g.beta = 0.f;
    const int bnt = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
    g.m_blocks = (int)avsi_ceil_div(g.M, 128), g.n_blocks = (int)avsi_ceil_div(Cout, bnt);
    g.k_split_len = (int)avsi_round_up(avsi_ceil_div(g.K, splits), 16);
    g.c_split_stride = M64 * ldo, g.n_group = g.n_blocks;
    g.conv_s0 = src0, g.conv_s1 = src1_coarse, g.conv_zeros = zeros64;
    g.cH = H, g.cW = W, g.cC0 = C0, g.cld0 = ld0, g.cC1 = C1, g.cld1 = ld1, g.ck = k;
    if ((int64_t)g.m_blocks * g.n_blocks > INT32_MAX) return AVSI_ERR_UNSUPPORTED;
    avsi_clear_error();
    const dim3 grid(g.m_blocks * g.n_blocks, 1, splits), block(256);
    const hipStream_t st = (hipStream_t)stream;
    if (bnt == 32)
        hipLaunchKernelGGL((gemm_dma_kernel<false, false, 16, 3, true, 32>), grid, block, (size_t)3 * (128 + 32) * 16 * 4, st, g);
    else if (bnt == 64)
        hipLaunchKernelGGL((gemm_dma_kernel<false, false, 16, 3, true, 64>), grid, block, (size_t)3 * (128 + 64) * 16 * 4, st, g);
    else
        hipLaunchKernelGGL((gemm_dma_kernel<false, false, 16, 3, true, 128>), grid, block, (size_t)3 * (128 + 128) * 16 * 4, st, g);
    if (avsi_launch_status() != AVSI_OK) return AVSI_ERR_LAUNCH;
    // every split writes its whole slab (columns < Cout of every row; an empty chunk writes zeros / the bias)
    return avsi_sum_slabs_launch(static_cast<const float*>(workspace), M64 * ldo, splits, M64 * ldo, out, 1.f, st);
}

int avsi_sum_slabs_launch(const float* slabs, int64_t n, int count, int64_t stride, float* out, float alpha,
                          hipStream_t st);

// Filter gradient of the convolution as an implicit GEMM: dW[(tap, c)][n] = sum over pixels of
// in(b, h+dh, w+dw, c) dY[(b,h,w)][n], i.e. im2col^T . dY without the im2col matrix; the reduction over
// the B*H*W pixels is cut into `splits` slabs summed in order (deterministic), like avsi_gemm_splitk_f32.
int avsi_sum_slabs_launch(const float* slabs, int64_t n, int count, int64_t stride, float* out, float alpha,
                          hipStream_t st);

extern "C" size_t avsi_conv2d_wgrad_workspace_bytes(int C0, int C1, int k, int Cout, int splits) {
    if (splits < 1 || k < 1 || Cout <= 0 || C0 + C1 <= 0) return 0;
    return (size_t)splits * (size_t)(k * k * (C0 + C1)) * (size_t)((Cout + 3) & ~3) * sizeof(float);
}

extern "C" int avsi_conv2d_wgrad_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B,
                                     int H, int W, int k, const float* dy, int ldy, int Cout, float* dw, int ldw,
                                     int splits, const float* zeros64, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!dy || !dw || !zeros64 || B <= 0 || H <= 0 || W <= 0 || k < 1 || !(k & 1) || Cout <= 0 || C0 < 0 || C1 < 0 || splits < 1)
        return AVSI_ERR_INVALID_ARG;
    if ((C0 && !src0) || (C1 && !src1_coarse) || ldy < Cout || ldw != ((Cout + 3) & ~3) || (C0 && ld0 < C0) || (C1 && ld1 < C1))
        return AVSI_ERR_INVALID_ARG;
    // a lane's 4 consecutive columns must stay inside one tap of one source; 16-byte gathers
    if ((C0 & 3) || (C1 & 3) || C0 + C1 < 4 || (ld0 & 3) || (ld1 & 3) || (ldy & 3) || (C1 && ((H | W) & 1)))
        return AVSI_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(src0) | reinterpret_cast<uintptr_t>(src1_coarse) | reinterpret_cast<uintptr_t>(dy) |
         reinterpret_cast<uintptr_t>(zeros64)) & 15)
        return AVSI_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < avsi_conv2d_wgrad_workspace_bytes(C0, C1, k, Cout, splits)) return AVSI_ERR_WORKSPACE;
    const int64_t R = (int64_t)B * H * W;
    if (R > INT32_MAX || (R & 15)) return AVSI_ERR_UNSUPPORTED;     // whole 16-pixel reduction tiles
    const int Kc = k * k * (C0 + C1);
    GemmArgs g{};
    g.A = zeros64, g.B = dy, g.C = (float*)workspace, g.bias = nullptr, g.row_scale = nullptr;
    g.M = Kc, g.N = Cout, g.K = (int)R;
    g.lda = 0, g.ldb = ldy, g.ldc = ldw;
    g.alpha = 1.f, g.beta = 0.f;
    const int bnt = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
    g.m_blocks = (int)avsi_ceil_div(Kc, 128), g.n_blocks = (int)avsi_ceil_div(Cout, bnt);
    g.k_split_len = (int)avsi_round_up(avsi_ceil_div(R, splits), 32);
    g.c_split_stride = (int64_t)Kc * ldw;
    g.n_group = g.n_blocks;
    g.conv_s0 = src0, g.conv_s1 = src1_coarse, g.conv_zeros = zeros64;
    g.cH = H, g.cW = W, g.cC0 = C0, g.cld0 = ld0, g.cC1 = C1, g.cld1 = ld1, g.ck = k;
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    const dim3 grid(g.m_blocks * g.n_blocks, 1, splits), block(256);
    if (bnt == 32)
        hipLaunchKernelGGL((gemm_dma_kernel<true, false, 16, 3, true, 32>), grid, block, (size_t)3 * (128 + 32) * 16 * 4, st, g);
    else if (bnt == 64)
        hipLaunchKernelGGL((gemm_dma_kernel<true, false, 16, 3, true, 64>), grid, block, (size_t)3 * (128 + 64) * 16 * 4, st, g);
    else
        hipLaunchKernelGGL((gemm_dma_kernel<true, false, 16, 3, true, 128>), grid, block, (size_t)3 * (128 + 128) * 16 * 4, st, g);
    const int rc = avsi_launch_status();
    if (rc != AVSI_OK) return rc;
    return avsi_sum_slabs_launch((const float*)workspace, (int64_t)Kc * ldw, splits, (int64_t)Kc * ldw, dw, 1.f, st);
}

extern "C" int avsi_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float beta, float* C, int64_t ldc,
                             const avsi_gemm_epilogue* epilogue, void* stream) {
    return avsi_gemm_launch(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, epilogue, 1, 0,
                            (hipStream_t)stream);
}

// ---- split-K: the reduction dimension is cut into `splits` contiguous chunks, one partial slab
// per chunk (plain stores, no atomics), summed in chunk order by a second kernel: deterministic.
// Used for the weight gradients, whose reduction runs over all T * Bp rows while the output is
// only a few hundred tiles.
int avsi_sum_slabs_launch(const float* slabs, int64_t n, int count, int64_t stride, float* out, float alpha,
                          hipStream_t st);

extern "C" size_t avsi_gemm_splitk_workspace_bytes(int M, int N, int splits) {
    if (M <= 0 || N <= 0 || splits < 1) return 0;
    return (size_t)splits * (size_t)M * (size_t)N * sizeof(float);
}

extern "C" int avsi_gemm_splitk_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A,
                                    int64_t lda, const float* B, int64_t ldb, float* C, int splits, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (splits < 1 || !C) return AVSI_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < avsi_gemm_splitk_workspace_bytes(M, N, splits)) return AVSI_ERR_WORKSPACE;
    const hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const int rc = avsi_gemm_launch(transA, transB, M, N, K, 1.f, A, lda, B, ldb, 0.f, ws, N, nullptr, splits,
                                    (int64_t)M * N, st);
    if (rc != AVSI_OK) return rc;
    // every split writes its whole slab (empty chunks write zeros), so the sum needs no memset
    return avsi_sum_slabs_launch(ws, (int64_t)M * N, splits, (int64_t)M * N, C, alpha, st);
}
