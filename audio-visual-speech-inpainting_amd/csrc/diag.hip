// Diagnostic entry points (no arithmetic of the hot path).
#include "avsi_common.h"

namespace {

// One workgroup per compute unit: 160 KiB of LDS leaves room for nothing else on the CU.  Waits (sleeping) until
// *release != 0 or the time budget is spent -- every workgroup leaves by itself, whatever the host does.
__global__ __launch_bounds__(64) void occupy_cu_kernel(const int* release, long long max_ticks) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) lds[0] = 1;       // keep the allocation
    const long long t0 = wall_clock64();    // constant 100 MHz counter
    while (true) {
        if (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        if (wall_clock64() - t0 > max_ticks) break;
        __builtin_amdgcn_s_sleep(127);
    }
}

// one wave that does nothing for `ticks` of the 100 MHz wall clock
__global__ __launch_bounds__(64) void stream_delay_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace

// Holds `stream` back for `microseconds` (<= 1000): one idle wave.  The trainer puts it in front of the weight-gradient
// GEMMs of a side stream that become ready at the same instant as a cooperative recurrent kernel on the main stream:
// dispatched first, their workgroups took every CU and the cooperative grid waited ~250 us for residency.
extern "C" int avsi_stream_delay_us(int microseconds, void* stream) {
    if (microseconds < 0 || microseconds > 1000) return AVSI_ERR_INVALID_ARG;
    if (microseconds == 0) return AVSI_OK;
    avsi_clear_error();
    hipLaunchKernelGGL(stream_delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100LL);
    return avsi_launch_status();
}

extern "C" int avsi_diag_occupy_cus(int num_cus, const int* release, int max_ms, void* stream) {
    if (num_cus <= 0 || num_cus > AVSI_NUM_CU || !release || max_ms <= 0 || max_ms > 60000) return AVSI_ERR_INVALID_ARG;
    constexpr int lds = 160 * 1024;
    avsi_clear_error();
    (void)hipFuncSetAttribute((const void*)occupy_cu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(occupy_cu_kernel, dim3(num_cus), dim3(64), lds, (hipStream_t)stream, release, (long long)max_ms * 100000LL);
    return avsi_launch_status();
}
