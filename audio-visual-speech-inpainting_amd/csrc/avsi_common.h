// Shared host-side helpers of the gfx950 hot-path library (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/avsi_hip.h"

#define AVSI_ABI_VERSION 9   // 9: step guard + guarded Adam (8: avsi_gemm_epilogue::k_zero (7: skewed-frame LWS sweeps (6: row-range forms of the cooperative forward entries (5: column-split recurrent kernel (4: LWS phase reconstruction (3: CTC loss + beam-search decoder (2: cooperative recurrence, implicit-GEMM / thin convolutions, blend loss)))))))

// Launch-status helpers.  hipGetLastError() is per-thread and also reports errors left behind
// by OTHER users of the runtime in this thread (e.g. the caller's framework), so every entry
// point clears it before its first launch and reads it back after its last.
static inline void avsi_clear_error() { (void)hipGetLastError(); }
static inline int avsi_launch_status() {
    return hipGetLastError() == hipSuccess ? AVSI_OK : AVSI_ERR_LAUNCH;
}

static inline int64_t avsi_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t avsi_round_up(int64_t a, int64_t b) { return avsi_ceil_div(a, b) * b; }

// MI355X: 256 CUs in 8 XCDs.  Used only to size persistent grids.
#define AVSI_NUM_CU 256
#define AVSI_NUM_XCD 8

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)), i.e. it waits for every global load AND store the wave still has in
// flight -- which is exactly the latency the pipelined loops keep in flight on purpose
// (tools/mfma_f32_lds.hip: 155 -> 56-89 TFLOP/s with four in-flight loads per barrier).
#define AVSI_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
