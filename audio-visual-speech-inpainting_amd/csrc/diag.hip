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

// Streaming copy, the yardstick of the HBM-bound kernels: 16 bytes per lane and access, four accesses in flight per lane,
// a grid sized to the chip (8 workgroups of 256 per CU) walking the buffer with a grid stride.
__global__ __launch_bounds__(256) void copy4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a, dst[i + stride] = b, dst[i + 2 * stride] = c, dst[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace

// dst[0 .. n) = src[0 .. n) (n a multiple of 4, both 16-byte aligned): bench.py's `device_copy_GB/s`.
extern "C" int avsi_diag_copy_f32(const float* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0 || (n & 3) || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15))
        return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    hipLaunchKernelGGL(copy4_kernel, dim3(AVSI_NUM_CU * 8), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), (long long)(n >> 2));
    return avsi_launch_status();
}

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
