// Issue rate of fp32 multiply-adds on gfx950 by operand form, at 1 / 2 / 4 waves per SIMD:
//   lit : v_fmac_f32 v, <32-bit literal>, v        (8-byte VOP2: the form the LWS sweep kernels use for their weights)
//   sgpr: v_fmac_f32 v, s, v                       (4-byte VOP2)
//   vgpr: v_fmac_f32 v, v, v                       (4-byte VOP2)
//   vop3: v_fma_f32 v, s, v, v                     (8-byte VOP3)
//   pk  : v_pk_fma_f32 v[2], s[2], v[2], v[2]      (8-byte VOP3P, two multiply-adds)
//   cnd32 / cnd64: v_cndmask_b32 with the mask in VCC (VOP2) / in an SGPR pair (VOP3);  mov: v_mov_b32 v, v;  xor: v_xor_b32 v, literal, v
// 16 independent accumulators, 256 instructions per loop pass.  hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int FORM>
__global__ __launch_bounds__(256) void k(float* out, int reps, float w0, float w1) {
    float a[16];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f b[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = v2f{a[2 * i], a[2 * i + 1]};
    const float x = threadIdx.x * 1e-6f;
    v2f xx = {x, x}, ww = {w0, w1};
    const unsigned long long mask = __builtin_amdgcn_read_exec() ^ (unsigned long long)(w0 > 0.f ? 0x5555 : 0);
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (FORM == 0) {
#define L(i) asm volatile("v_fmac_f32 %0, 0x3dc53288, %1" : "+v"(a[i]) : "v"(x));
                REP16(L)
#undef L
            } else if (FORM == 1) {
#define L(i) asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a[i]) : "v"(x), "s"(w0));
                REP16(L)
#undef L
            } else if (FORM == 2) {
#define L(i) asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a[i]) : "v"(x), "v"(a[(i + 5) & 15]));
                REP16(L)
#undef L
            } else if (FORM == 3) {
#define L(i) asm volatile("v_fma_f32 %0, %2, %1, %0" : "+v"(a[i]) : "v"(x), "s"(w0));
                REP16(L)
#undef L
            } else if (FORM == 4) {
#define L(i) asm volatile("v_pk_fma_f32 %0, %2, %1, %0" : "+v"(b[i & 7]) : "v"(xx), "s"(ww));
                REP16(L)
#undef L
            } else if (FORM == 5) {
                asm volatile("v_cmp_gt_f32 vcc, %16, %17\n\t"
                             "v_cndmask_b32_e32 %0, %0, %16, vcc\n\tv_cndmask_b32_e32 %1, %1, %16, vcc\n\tv_cndmask_b32_e32 %2, %2, %16, vcc\n\t"
                             "v_cndmask_b32_e32 %3, %3, %16, vcc\n\tv_cndmask_b32_e32 %4, %4, %16, vcc\n\tv_cndmask_b32_e32 %5, %5, %16, vcc\n\t"
                             "v_cndmask_b32_e32 %6, %6, %16, vcc\n\tv_cndmask_b32_e32 %7, %7, %16, vcc\n\tv_cndmask_b32_e32 %8, %8, %16, vcc\n\t"
                             "v_cndmask_b32_e32 %9, %9, %16, vcc\n\tv_cndmask_b32_e32 %10, %10, %16, vcc\n\tv_cndmask_b32_e32 %11, %11, %16, vcc\n\t"
                             "v_cndmask_b32_e32 %12, %12, %16, vcc\n\tv_cndmask_b32_e32 %13, %13, %16, vcc\n\tv_cndmask_b32_e32 %14, %14, %16, vcc\n\t"
                             "v_cndmask_b32_e32 %15, %15, %16, vcc"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                               "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                             : "v"(x), "v"(w0)
                             : "vcc");
            } else if (FORM == 6) {
#define L(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "s"(mask));
                REP16(L)
#undef L
            } else if (FORM == 7) {
#define L(i) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 5) & 15]));
                REP16(L)
#undef L
            } else if (FORM == 8) {
#define L(i) asm volatile("v_xor_b32_e32 %0, 0x80000000, %0" : "+v"(a[i]));
                REP16(L)
#undef L
            } else if (FORM == 9) {
#define L(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b[i & 7]) : "v"(xx), "v"(b[(i + 3) & 7]));
                REP16(L)
#undef L
            } else if (FORM == 10) {
#define L(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(b[i & 7]) : "v"(xx));
                REP16(L)
#undef L
            } else if (FORM == 11) {
#define L(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(b[i & 7]) : "v"(xx));
                REP16(L)
#undef L
            } else if (FORM == 12) {
#define L(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
                REP16(L)
#undef L
            } else {
#define L(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(a[(i + 5) & 15]));
                REP16(L)
#undef L
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += b[i].x + b[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int CU = 256, reps = 4000;
    float* out;
    if (hipMalloc(&out, (size_t)CU * 8 * 256 * 4) != hipSuccess) return 1;
    const char* names[14] = {"lit ", "sgpr", "vgpr", "vop3", "pk  ", "cnd32", "cnd64", "mov ", "xor ", "pkfma_vvv", "pkadd_vv", "pkmul_vv", "add_vv", "fma_vvv(vop3)"};
    for (int wps : {1, 2, 4}) {
        for (int f = 0; f < 14; ++f) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
            float ms = 0;
            for (int it = 0; it < 2; ++it) {
                (void)hipEventRecord(e0, 0);
                switch (f) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 6: hipLaunchKernelGGL(k<6>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 7: hipLaunchKernelGGL(k<7>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 8: hipLaunchKernelGGL(k<8>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 9: hipLaunchKernelGGL(k<9>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 10: hipLaunchKernelGGL(k<10>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 11: hipLaunchKernelGGL(k<11>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    case 12: hipLaunchKernelGGL(k<12>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                    default: hipLaunchKernelGGL(k<13>, dim3(CU * wps), dim3(256), 0, 0, out, reps, 0.1f, 0.2f); break;
                }
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double ns_per_inst_simd = ms * 1e6 / ((double)reps * 256.0 * wps);     // one SIMD runs wps waves
            printf("%d waves/SIMD %s: %.3f ns per wave-instruction and SIMD (%.2f cycles at 2.4 GHz)\n", wps, names[f], ns_per_inst_simd, ns_per_inst_simd * 2.4);
        }
    }
    return 0;
}
