// Ping-pong between two workgroups through global memory: which store / load / poll flavour is fastest, and is it
// coherent?  Blocks 0 and `stride` exchange a 128 KiB (or smaller) buffer per round.
// hipcc --offload-arch=gfx950 -O2 tools/xcd_pingpong.hip -o tools/xcd_pingpong && tools/xcd_pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int LOADK>
__device__ __forceinline__ v4f ld(const float* p) {
    v4f v;
    if (LOADK == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LOADK == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LOADK == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LOADK == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int K>
__device__ __forceinline__ void ld_issue(v4f& v, const float* p) {
    if (K == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (K == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (K == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (K == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
}
template <int K>
__device__ __forceinline__ void st(float* p, v4f v) {
    if (K == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (K == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (K == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (K == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int K>
__device__ __forceinline__ unsigned poll(unsigned* p) {
    unsigned v;
    if (K == 0) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (K == 1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (K == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (K == 3) asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(0u) : "memory");
    return v;
}
template <int K>
__device__ __forceinline__ void flag_inc(unsigned* p) {
    if (K == 0) asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(1u) : "memory");
    if (K == 1) asm volatile("global_atomic_add %0, %1, off sc1" ::"v"(p), "v"(1u) : "memory");
}

// NV = float4 per lane per round (512 lanes): bytes per round = NV * 512 * 16
template <int STK, int LDK, int POLLK, int INCK, int NV>
__global__ __launch_bounds__(512) void pingpong(float* buf, unsigned* flags, unsigned* out, int stride, int rounds) {
    extern __shared__ char smem[];
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == stride ? 1 : -1);
    if (me < 0) return;
    float* mine = buf + (size_t)me * NV * 512 * 4;
    float* theirs = buf + (size_t)(1 - me) * NV * 512 * 4;
    unsigned* myflag = flags + me * 64;
    unsigned* theirflag = flags + (1 - me) * 64;
    unsigned bad = 0, timeouts = 0;
    const long long t0 = clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (me == 1 || r > 1) {
            // wait for the peer's round (me == 0 starts)
            const unsigned want = me == 1 ? r : r - 1;
            if (threadIdx.x == 0) {
                unsigned polls = 0;
                while (poll<POLLK>(theirflag) < want)
                    if (++polls > (1u << 20)) { ++timeouts; break; }
            }
            __syncthreads();
            v4f v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) ld_issue<LDK>(v[i], theirs + (i * 512 + threadIdx.x) * 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                asm volatile("" : "+v"(v[i]));
                bad += v[i].x != (float)want;
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v4f x = {(float)r, (float)r, (float)r, (float)r};
            st<STK>(mine + (i * 512 + threadIdx.x) * 4, x);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) flag_inc<INCK>(myflag);
    }
    const long long t1 = clock64();
    atomicAdd(out + me * 4 + 1, bad);
    if (threadIdx.x == 0) {
        out[me * 4] = (unsigned)((t1 - t0) / rounds);
        out[me * 4 + 2] = timeouts;
    }
    if (threadIdx.x == 1000) smem[0] = 1;
}

template <int STK, int LDK, int POLLK, int INCK, int NV>
void run(const char* name, int stride) {
    float* buf;
    unsigned *flags, *out;
    hipMalloc(&buf, 2 * NV * 512 * 16);
    hipMalloc(&flags, 1024);
    hipMalloc(&out, 64);
    hipMemset(buf, 0, 2 * NV * 512 * 16);
    hipMemset(flags, 0, 1024);
    hipMemset(out, 0, 64);
    hipFuncSetAttribute((const void*)pingpong<STK, LDK, POLLK, INCK, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const int rounds = 200;
    hipLaunchKernelGGL((pingpong<STK, LDK, POLLK, INCK, NV>), dim3(stride + 1), dim3(512), 96 * 1024, 0, buf, flags, out, stride, rounds);
    unsigned h[8];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("%-34s stride %2d  %3d KiB: %6u clk/round (100 MHz clock64: %.2f us)  wrong values %u  timeouts %u\n", name, stride,
           NV * 8, h[0], h[0] / 100.0, h[1] + h[5], h[2] + h[6]);
    hipFree(buf);
    hipFree(flags);
    hipFree(out);
}

int main() {
    // same XCD (stride 8) with L2-scope flavours
    run<0, 1, 0, 0, 16>("st plain, ld sc0, poll ld sc0", 8);
    run<0, 1, 3, 0, 16>("st plain, ld sc0, poll rmw", 8);
    run<0, 0, 0, 0, 16>("st plain, ld plain, poll ld sc0", 8);
    run<0, 2, 1, 0, 16>("st plain, ld sc1, poll ld sc1", 8);
    run<2, 2, 1, 1, 16>("st sc1, ld sc1, poll sc1, inc sc1", 8);
    run<3, 3, 2, 1, 16>("st sc0sc1, ld sc0sc1, poll sc0sc1", 8);
    run<0, 1, 0, 0, 4>("st plain, ld sc0, poll ld sc0", 8);
    run<3, 3, 2, 1, 4>("st sc0sc1, ld sc0sc1, poll sc0sc1", 8);
    run<0, 1, 0, 0, 1>("st plain, ld sc0, poll ld sc0", 8);
    run<3, 3, 2, 1, 1>("st sc0sc1, ld sc0sc1, poll sc0sc1", 8);
    // different XCDs (stride 1): the L2-scope flavour must fail or be wrong, the device-scope one must work
    run<0, 1, 0, 0, 16>("st plain, ld sc0, poll ld sc0", 1);
    run<3, 3, 2, 1, 16>("st sc0sc1, ld sc0sc1, poll sc0sc1", 1);
    run<2, 2, 1, 1, 16>("st sc1, ld sc1, poll sc1, inc sc1", 1);
    return 0;
}
