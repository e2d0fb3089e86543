// Does a CU-masked stream (hipExtStreamCreateWithCUMask) confine a kernel to some XCDs, and how are its workgroups dealt to
// the XCDs then?  The question behind it: can LWS phase refinement be kept on XCDs 4 - 7 while the model step of a
// batch of 32 (whose cooperative recurrent kernels sit on XCDs 0 - 3) runs beside it.
//   hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/cu_mask_probe && timeout -k 5 60 tools/cu_mask_probe
// Mask bit j is CU j / 8 of XCC j % 8 (KFD deals the user's bits to the XCCs in turn).  No XCC is ever left without a CU here:
// a queue with an XCC that cannot run anything is not something to try on a shared box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <chrono>

__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) out[2 * blockIdx.x] = xcc & 0xF, out[2 * blockIdx.x + 1] = hwid;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(10);
}

static void run(const char* name, hipStream_t st, int blocks, int spin) {
    unsigned* d;
    hipMalloc(&d, blocks * 8);
    hipMemset(d, 0xFF, blocks * 8);
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, st, d, spin);
    hipError_t e = hipStreamSynchronize(st);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<unsigned> h(2 * blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    int per[16] = {0}, round_robin = 0;
    for (int i = 0; i < blocks; ++i) {
        per[h[2 * i] & 15]++;
        round_robin += (h[2 * i] & 15) == (unsigned)(i % 8);
    }
    printf("%-34s grid %4d: %s, %.2f ms; workgroups per XCC:", name, blocks, hipGetErrorString(e), ms);
    for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
    printf("; XCC_ID == blockIdx %% 8 for %d of %d\n", round_robin, blocks);
    hipFree(d);
}

int main() {
    hipStream_t plain, lo, hi;
    hipStreamCreate(&plain);
    // 256 CUs = 8 words; XCC of bit j = j % 8.  `lo`: every CU of XCCs 0 - 3 and CU 0 of XCCs 4 - 7; `hi`: the mirror image
    uint32_t mlo[8], mhi[8];
    for (int w = 0; w < 8; ++w) mlo[w] = 0x0F0F0F0Fu, mhi[w] = 0xF0F0F0F0u;
    mlo[0] |= 0xF0u, mhi[0] |= 0x0Fu;
    hipError_t e1 = hipExtStreamCreateWithCUMask(&lo, 8, mlo), e2 = hipExtStreamCreateWithCUMask(&hi, 8, mhi);
    printf("hipExtStreamCreateWithCUMask: %s, %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    if (e1 != hipSuccess || e2 != hipSuccess) return 1;
    for (int blocks : {64, 256, 1024}) {
        run("no mask", plain, blocks, 200);
        run("XCCs 0-3 (+ one CU of each other)", lo, blocks, 200);
        run("XCCs 4-7 (+ one CU of each other)", hi, blocks, 200);
    }
    // both at once: does a kernel on `hi` keep off the CUs of `lo`?  (long-running on hi, then a timed short one on lo)
    unsigned* d;
    hipMalloc(&d, 2048 * 8);
    hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, hi, d, 20000);
    run("XCCs 0-3 beside a busy 4-7 stream", lo, 256, 200);
    run("no mask beside a busy 4-7 stream", plain, 256, 200);
    hipDeviceSynchronize();
    hipFree(d);
    return 0;
}
