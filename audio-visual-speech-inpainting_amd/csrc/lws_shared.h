// Pieces of lws.hip that lws_skew.hip uses as well (internal to the library, not part of the C ABI).
#pragma once
#include "avsi_common.h"

constexpr int AVSI_LWS_NF = 512;             // FFT length the kernels are written for
constexpr int AVSI_LWS_KB = AVSI_LWS_NF / 2 + 1;
constexpr int AVSI_LWS_LMAX = 5;
constexpr int AVSI_LWS_NP = 2 * AVSI_LWS_LMAX + 1;
constexpr int AVSI_LWS_MAX_SWEEPS = 256;

struct AvsiLwsSchedule {
    int n;
    float rel[AVSI_LWS_MAX_SWEEPS];            // threshold relative to the mean magnitude of the utterance
    unsigned char past_only[AVSI_LWS_MAX_SWEEPS];
};

// alpha_q(p), q = -1, 0, +1 (rows |q| >= 2 vanish: frame_len <= 2 hop), p = -LMAX .. LMAX, zero beyond |p| > L; double
void avsi_lws_host_alpha(int frame_len, int hop, int nfft, int L, double alpha[3][AVSI_LWS_NP][2]);
bool avsi_lws_geometry_ok(int frame_len, int hop, int nfft);
// the sweeps of one run_lws call (oracle/lws.py LWS.sweep_schedule); false if there are more than MAX_SWEEPS
bool avsi_lws_make_schedule(int nofuture_iterations, float nofuture_alpha, int online_iterations, float online_alpha,
                            int batch_iterations, float batch_alpha, float batch_beta, float batch_gamma, AvsiLwsSchedule& S);
// stats[b] = (mean, max) of |spec[b]| ([M][257] complex each)
void avsi_lws_launch_stats(const float* spec, int batch, int M, float* stats, hipStream_t st);
