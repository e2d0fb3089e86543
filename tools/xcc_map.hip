// Which XCD does block i land on?  hipcc --offload-arch=gfx950 -O2 tools/xcc_map.hip -o tools/xcc_map && tools/xcc_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(512) void probe(unsigned* out) {
    extern __shared__ char smem[];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hwid;
    }
    // stay resident for a while so that the whole grid is co-resident (one block per CU)
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 1000) smem[0] = 1;
}

int main() {
    for (int blocks : {32, 64, 256, 264}) {
        unsigned* d;
        hipMalloc(&d, blocks * 8);
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 96 * 1024, 0, d);
        std::vector<unsigned> h(2 * blocks);
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < blocks; ++i) bad += ((h[2 * i] & 0xF) != (unsigned)(i % 8));
        printf("grid %d: blocks whose XCC_ID != blockIdx %% 8: %d;  first 16 xcc ids:", blocks, bad);
        for (int i = 0; i < 16 && i < blocks; ++i) printf(" %u", h[2 * i] & 0xF);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
