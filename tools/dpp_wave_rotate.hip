#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* o, const float* in) {
    float v = in[threadIdx.x];
    float a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false));  // wave_ror:1
    float b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x134, 0xf, 0xf, false));  // wave_rol:1
    asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
    o[threadIdx.x] = a;
    o[64 + threadIdx.x] = b;
}
int main() {
    float h[64], r[128], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = i;
    hipMalloc(&di, 256); hipMalloc(&dout, 512);
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    hipMemcpy(r, dout, 512, hipMemcpyDeviceToHost);
    printf("ror: lane0 <- %g, lane1 <- %g, lane63 <- %g\n", r[0], r[1], r[63]);
    printf("rol: lane0 <- %g, lane1 <- %g, lane63 <- %g\n", r[64], r[65], r[127]);
    return 0;
}
