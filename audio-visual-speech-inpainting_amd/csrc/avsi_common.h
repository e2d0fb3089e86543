// Shared host-side helpers of the gfx950 hot-path library (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/avsi_hip.h"

#define AVSI_ABI_VERSION 12  // 12: avsi_sgd_momentum_f32 (11: avsi_blstm_rec_bwd_kernel_name (10: two-utterances-per-wave LWS sweeps) (9: step guard + guarded Adam (8: avsi_gemm_epilogue::k_zero (7: skewed-frame LWS sweeps (6: row-range forms of the cooperative forward entries (5: column-split recurrent kernel (4: LWS phase reconstruction (3: CTC loss + beam-search decoder (2: cooperative recurrence, implicit-GEMM / thin convolutions, blend loss)))))))))

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

// Bounded wait of the cooperative recurrent kernels, in WALL-CLOCK time.  A member normally sees its peers within
// microseconds; at the start of a launch it may wait for them to become resident behind whatever else occupies the
// chip (milliseconds).  Beyond AVSI_COOP_TIMEOUT_MS (default 2000, 10 .. 60000) the launch is taken to be
// over-subscribed -- another resident of the GPU left its workgroups no room -- and gives up (sticky status word).  A
// poll COUNT was used before: 2^22 polls turned out to be more than 30 s once sleeping polls of 30+ workgroups queue on
// one memory channel.  The clock (s_memrealtime, 100 MHz) is read on every 128th poll only.
static inline long long avsi_coop_spin_ticks() {
    static long long ticks = 0;
    if (!ticks) {
        const char* e = getenv("AVSI_COOP_TIMEOUT_MS");
        long long ms = e ? atoll(e) : 2000;
        ms = ms < 10 ? 10 : (ms > 60000 ? 60000 : ms);
        ticks = ms * 100000LL;
    }
    return ticks;
}
#if defined(__HIPCC__)
__device__ __forceinline__ bool avsi_spin_expired(unsigned& polls, long long& t0, long long limit_ticks,
                                                  const unsigned* status) {
    if ((++polls & 127u) != 0) return false;
    // somebody else of this launch (or of one before it) has given up already: everything is void, do not wait out a
    // bound of one's own behind it (members that become resident one after the other would otherwise chain their waits)
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    const long long now = wall_clock64();
    if (polls == 128u) {
        t0 = now;
        return false;
    }
    return now - t0 > limit_ticks;
}

// The sticky status word as ONE verdict per workgroup (thread 0 reads it, every thread branches on that value): read per
// thread, a word that flips between the reads of two waves would send some waves of the workgroup into its barriers and
// LDS exchanges and let the others leave.  One barrier, once per launch.
__device__ __forceinline__ bool avsi_launch_is_void(const unsigned* status) {
    __shared__ unsigned verdict;
    if (threadIdx.x == 0) verdict = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return verdict != 0;
}
#endif
