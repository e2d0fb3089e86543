// Diagnostic harness: phase breakdown (cycles) of one step of the recurrent BLSTM kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAVSI_REC_STAMPS -o rec_stamps rec_stamps.cpp \
//        ../audio-visual-speech-inpainting_amd/csrc/blstm_fwd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" int avsi_blstm_rec_fwd_stamps(const float*, const float*, float*, int, int, int, unsigned long long*, void*);
int main(int argc, char** argv) {
    const int T = 50, Bp = argc > 1 ? atoi(argv[1]) : 8192, mt = argc > 2 ? atoi(argv[2]) : 64;
    const size_t nx = (size_t)T * Bp * 2048, nh = (size_t)T * Bp * 512, nw = 2 * 262144;
    float *x, *w, *h;
    unsigned long long* st;
    (void)hipMalloc(&x, nx * 4), (void)hipMalloc(&w, nw * 4), (void)hipMalloc(&h, nh * 4);
    const int wgs = 2 * ((Bp + mt - 1) / mt);
    (void)hipMalloc(&st, (size_t)wgs * 8 * 4 * 8);
    std::vector<float> hx(1 << 20), hw(nw);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (size_t o = 0; o < nx; o += hx.size()) (void)hipMemcpy(x + o, hx.data(), std::min(hx.size(), nx - o) * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        int rc = avsi_blstm_rec_fwd_stamps(x, w, h, T, Bp, mt, st, nullptr);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hs((size_t)wgs * 8 * 4);
        (void)hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
        double ph[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < hs.size(); ++i) ph[i & 3] += (double)hs[i];
        const double n = (double)wgs * 8 * T;
        printf("rc=%d Bp=%d mt=%d: %.2f ms total, %.1f us/step | per step per wave (s_memtime ticks): load %.0f  mfma %.0f  cell %.0f  barrier %.0f  sum %.0f\n",
               rc, Bp, mt, ms, ms * 1e3 / T, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, (ph[0] + ph[1] + ph[2] + ph[3]) / n);
    }
    return 0;
}
