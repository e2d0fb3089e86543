// Streaming-copy yardstick on MI355X: what does a plain device copy reach, by buffer size, grid, unroll and cache policy?
// (MI355X_MICROARCH.md quotes ~6.3 TB/s achievable; avsi_diag_copy_f32 measured 4.6 - 5.1 TB/s on this pool's boxes.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/copy_sweep tools/copy_sweep.hip && tools/copy_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int UNR, int NT>
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256 * UNR;
    for (size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x; i < n4; i += stride) {
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < n4) {
                if (NT & 1) {
                    const float* p = reinterpret_cast<const float*>(src + j);
                    v[u].x = __builtin_nontemporal_load(p), v[u].y = __builtin_nontemporal_load(p + 1);
                    v[u].z = __builtin_nontemporal_load(p + 2), v[u].w = __builtin_nontemporal_load(p + 3);
                } else {
                    v[u] = src[j];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < n4) {
                if (NT & 2) {
                    float* p = reinterpret_cast<float*>(dst + j);
                    __builtin_nontemporal_store(v[u].x, p), __builtin_nontemporal_store(v[u].y, p + 1);
                    __builtin_nontemporal_store(v[u].z, p + 2), __builtin_nontemporal_store(v[u].w, p + 3);
                } else {
                    dst[j] = v[u];
                }
            }
        }
    }
}

__global__ void read_kernel(const float4* __restrict__ src, float* __restrict__ out, size_t n4) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}

__global__ void write_kernel(float4* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const size_t max_bytes = (size_t)8 << 30;
    float4 *src, *dst;
    float* out;
    hipMalloc(&src, max_bytes), hipMalloc(&dst, max_bytes), hipMalloc(&out, 4);
    hipMemset(src, 1, max_bytes), hipMemset(dst, 0, max_bytes);
    for (size_t mib : {256, 1024, 4096, 8192}) {
        const size_t n4 = (mib << 20) / 16;
        for (int wg_per_cu : {4, 8, 16, 64}) {
            const int grid = 256 * wg_per_cu;
#define RUN(UNR, NT)                                                                                                        \
    {                                                                                                                       \
        const double ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<UNR, NT>), dim3(grid), dim3(256), 0, 0, src, dst, n4); }, 5); \
        printf("copy %5zu MiB grid %5d unroll %d nt %d: %.3f ms %.0f GB/s (read + write)\n", mib, grid, UNR, NT, ms,       \
               2.0 * (mib << 20) / ms / 1e6);                                                                               \
    }
            RUN(1, 0) RUN(4, 0) RUN(8, 0) RUN(4, 3) RUN(4, 2) RUN(4, 1)
#undef RUN
        }
        const double mr = time_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(256 * 16), dim3(256), 0, 0, src, out, n4); }, 5);
        const double mw = time_ms([&] { hipLaunchKernelGGL(write_kernel, dim3(256 * 16), dim3(256), 0, 0, dst, n4); }, 5);
        const double mm = time_ms([&] { hipMemcpyAsync(dst, src, mib << 20, hipMemcpyDeviceToDevice, 0); }, 5);
        printf("read-only %5zu MiB: %.3f ms %.0f GB/s; write-only: %.3f ms %.0f GB/s; hipMemcpyAsync D2D: %.3f ms %.0f GB/s (read + write)\n",
               mib, mr, (mib << 20) / mr / 1e6, mw, (mib << 20) / mw / 1e6, mm, 2.0 * (mib << 20) / mm / 1e6);
        fflush(stdout);
    }
    return 0;
}
