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
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void copy4_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, long long n4) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    auto ld = [&](long long j) { return NT ? __builtin_nontemporal_load(src + j) : src[j]; };
    auto st = [&](long long j, v4f v) {
        if (NT) __builtin_nontemporal_store(v, dst + j);
        else dst[j] = v;
    };
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const v4f a = ld(i), b = ld(i + stride), c = ld(i + 2 * stride), d = ld(i + 3 * stride);
        st(i, a), st(i + stride, b), st(i + 2 * stride, c), st(i + 3 * stride, d);
    }
    for (; i < n4; i += stride) st(i, ld(i));
}

}  // namespace

// dst[0 .. n) = src[0 .. n) (n a multiple of 4, both 16-byte aligned): bench.py's `device_copy_GB/s`.
// AVSI_DIAG_COPY_NT=1: non-temporal loads and stores.
extern "C" int avsi_diag_copy_f32(const float* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0 || (n & 3) || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15))
        return AVSI_ERR_INVALID_ARG;
    avsi_clear_error();
    const char* e = getenv("AVSI_DIAG_COPY_NT");
    if (e && e[0] == '1')
        hipLaunchKernelGGL(copy4_kernel<true>, dim3(AVSI_NUM_CU * 8), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const v4f*>(src), reinterpret_cast<v4f*>(dst), (long long)(n >> 2));
    else
        hipLaunchKernelGGL(copy4_kernel<false>, dim3(AVSI_NUM_CU * 8), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const v4f*>(src), reinterpret_cast<v4f*>(dst), (long long)(n >> 2));
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
