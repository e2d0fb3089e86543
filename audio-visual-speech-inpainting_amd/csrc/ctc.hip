// CTC head of the multi-task models (reference models.py:1934-1952, 2026-2031): the loss
// tf.nn.ctc_loss(labels, logits, sequence_length) with its gradient w.r.t. the un-normalised logits on
// the GPU, and the beam-search decoder behind the phone-error-rate diagnostic on the host (TensorFlow
// runs that decoder on the CPU as well; it is bookkeeping over 20 label prefixes, not arithmetic).
//
// Loss kernel: one workgroup per utterance.
//   1. all four waves: per-frame log-sum-exp, then the table lp[t][s] = log softmax(logits[t])[l'_s] of the
//      blank-extended labelling l' (S = 2L + 1 states) into LDS -- the only pass that gathers from HBM;
//   2. wave 0 runs the forward recursion (alpha), wave 1 the backward one (beta) at the same time.  A lane
//      owns NS consecutive states in registers, the two neighbours come from the adjacent lane by
//      shuffles, so a step is LDS reads + ALU with no barrier; both tables go to the workspace.  Every
//      fourth step the wave subtracts its largest logarithm (the sum of these offsets, kept in double,
//      rebuilds log p): the stored logarithms stay small, so float32 resolves them to ~1e-7 instead of
//      the 6e-5 ulp of the raw magnitudes (several hundred after 250 frames);
//   3. all four waves: per frame gamma[s] = softmax_s(alpha + beta - lp) -- the state posteriors of a
//      frame sum to one, so each frame is normalised by its own sum and no offset is needed -- summed
//      per class in a fixed order (no atomics: results are reproducible), grad = scale * (softmax - sum).
// Latency-bound by design (T dependent steps); every utterance of the batch runs concurrently.
#include <algorithm>
#include <cmath>
#include <vector>

#include "avsi_common.h"

namespace {

constexpr float NEG = -1e30f;        // log(0): absorbs every addition the recursions make
constexpr int CTPB = 256;

__device__ __forceinline__ float clampneg(float v) { return v < -1e29f ? NEG : v; }

__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m < -1e29f) return NEG;
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// subtracts the wave's largest logarithm from the NS states of every lane; returns it (0 if all are log 0)
template <int NS>
__device__ __forceinline__ float renorm(float (&a)[NS]) {
    float m = a[0];
#pragma unroll
    for (int i = 1; i < NS; ++i) m = fmaxf(m, a[i]);
    m = wave_max(m);
    if (m < -1e29f) return 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) a[i] = clampneg(a[i] - m);
    return m;
}

template <int NS>
__global__ __launch_bounds__(CTPB) void ctc_loss_kernel(const float* __restrict__ logits, int64_t ld_b, int64_t ld_t,
                                                        int T, int C, const int32_t* __restrict__ labels,
                                                        int label_pitch, const int32_t* __restrict__ label_len,
                                                        const int32_t* __restrict__ seq_len, int max_label_len,
                                                        float gscale, float* __restrict__ loss,
                                                        float* __restrict__ grad, float* __restrict__ ws) {
    constexpr int SC = 64 * NS;          // state capacity = row pitch of the alpha / beta tables
    extern __shared__ float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(max(seq_len[b], 0), T);
    const int L = min(max(label_len[b], 0), max_label_len);
    const int S = 2 * L + 1, blank = C - 1;
    __shared__ double s_offset;                          // sum of the offsets taken out of alpha
    float* s_lse = smem;                                 // [T]
    int* s_ext = reinterpret_cast<int*>(smem + T);       // [SC]
    float* s_gam = smem + T + SC;                        // [4][SC]
    float* s_lp = s_gam + 4 * SC;                        // [n][S]
    const float* lg = logits + (int64_t)b * ld_b;
    float* al = ws + (size_t)b * 2 * T * SC;
    float* be = al + (size_t)T * SC;

    for (int s = tid; s < SC; s += CTPB)
        s_ext[s] = (s < S && (s & 1)) ? min(max(labels[(int64_t)b * label_pitch + (s >> 1)], 0), blank) : blank;
    for (int t = wave; t < n; t += CTPB / 64) {
        float m = -INFINITY;
        for (int k = lane; k < C; k += 64) m = fmaxf(m, lg[t * ld_t + k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float e = 0.f;
        for (int k = lane; k < C; k += 64) e += __expf(lg[t * ld_t + k] - m);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
        if (lane == 0) s_lse[t] = m + __logf(e);
    }
    __syncthreads();
    for (int idx = tid; idx < n * S; idx += CTPB) {
        const int t = idx / S, s = idx - t * S;
        s_lp[idx] = lg[t * ld_t + s_ext[s]] - s_lse[t];
    }
    __syncthreads();

    if (wave == 0 && n > 0) {
        double offset = 0.0;
        float a[NS];
        bool valid[NS], skip[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int s = lane * NS + i;
            valid[i] = s < S;
            skip[i] = valid[i] && s >= 2 && (s & 1) && s_ext[s] != s_ext[s - 2];
            a[i] = (valid[i] && s < 2) ? s_lp[s] : NEG;
            al[s] = a[i];
        }
        for (int t = 1; t < n; ++t) {
            float up1 = __shfl_up(a[NS - 1], 1, 64);
            float up2 = NS >= 2 ? __shfl_up(a[NS >= 2 ? NS - 2 : 0], 1, 64) : __shfl_up(a[0], 2, 64);
            if (lane < 1) up1 = NEG;
            if (lane < (NS >= 2 ? 1 : 2)) up2 = NEG;
            float nw[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const float p1 = i >= 1 ? a[i >= 1 ? i - 1 : 0] : up1;
                const float p2 = i >= 2 ? a[i >= 2 ? i - 2 : 0] : (i == 1 ? up1 : up2);
                const int s = lane * NS + i;
                nw[i] = valid[i] ? clampneg(lse3(a[i], p1, skip[i] ? p2 : NEG) + s_lp[t * S + s]) : NEG;
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) a[i] = nw[i];
            if ((t & 3) == 3) offset += (double)renorm<NS>(a);
#pragma unroll
            for (int i = 0; i < NS; ++i) al[(size_t)t * SC + lane * NS + i] = a[i];
        }
        if (lane == 0) s_offset = offset;
    } else if (wave == 1 && n > 0) {
        float v[NS];
        bool valid[NS], skip[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int s = lane * NS + i;
            valid[i] = s < S;
            skip[i] = s + 2 < S && ((s + 2) & 1) && s_ext[s + 2] != s_ext[s];
            v[i] = (valid[i] && s >= S - 2) ? s_lp[(n - 1) * S + s] : NEG;
            be[(size_t)(n - 1) * SC + s] = v[i];
        }
        for (int t = n - 2; t >= 0; --t) {
            float dn1 = __shfl_down(v[0], 1, 64);
            float dn2 = NS >= 2 ? __shfl_down(v[NS >= 2 ? 1 : 0], 1, 64) : __shfl_down(v[0], 2, 64);
            if (lane > 62) dn1 = NEG;
            if (lane > (NS >= 2 ? 62 : 61)) dn2 = NEG;
            float nw[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const float n1 = i + 1 < NS ? v[i + 1 < NS ? i + 1 : 0] : dn1;
                const float n2 = i + 2 < NS ? v[i + 2 < NS ? i + 2 : 0] : (i + 1 < NS ? dn1 : dn2);
                const int s = lane * NS + i;
                nw[i] = valid[i] ? clampneg(lse3(v[i], n1, skip[i] ? n2 : NEG) + s_lp[t * S + s]) : NEG;
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) v[i] = nw[i];
            if ((t & 3) == 0) (void)renorm<NS>(v);
#pragma unroll
            for (int i = 0; i < NS; ++i) be[(size_t)t * SC + lane * NS + i] = v[i];
        }
    }
    __syncthreads();        // drains the table stores of waves 0 and 1 (s_waitcnt vmcnt(0)) before anyone reads them

    float log_p = NEG;
    if (n > 0) {
        const float x = al[(size_t)(n - 1) * SC + S - 1], y = S > 1 ? al[(size_t)(n - 1) * SC + S - 2] : NEG;
        log_p = lse3(x, y, NEG);
        if (log_p > -1e29f) log_p = (float)((double)log_p + s_offset);
    } else if (L == 0) {
        log_p = 0.f;
    }
    const bool feasible = log_p > -1e29f;
    if (tid == 0) loss[b] = feasible ? -log_p : INFINITY;
    if (!grad) return;
    float* gr = grad + (int64_t)b * ld_b;
    float* gam = s_gam + wave * SC;
    for (int t = wave; t < T; t += CTPB / 64) {
        if (t >= n || !feasible) {
            for (int k = lane; k < C; k += 64) gr[t * ld_t + k] = 0.f;
            continue;
        }
        float gmax = NEG;
        for (int s = lane; s < S; s += 64) {
            const float g = clampneg(al[(size_t)t * SC + s] + be[(size_t)t * SC + s] - s_lp[t * S + s]);
            gam[s] = g;
            gmax = fmaxf(gmax, g);
        }
        gmax = wave_max(gmax);
        float gsum = 0.f;
        for (int s = lane; s < S; s += 64) {
            const float e = __expf(gam[s] - gmax);          // own element: no cross-lane dependence yet
            gam[s] = e;
            gsum += e;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gsum += __shfl_xor(gsum, o, 64);
        const float ginv = 1.f / gsum;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same-wave LDS write -> read
        for (int k = lane; k < C; k += 64) {
            float acc = 0.f;
            if (k == blank) {
                for (int j = 0; j <= L; ++j) acc += gam[2 * j];
            } else {
                for (int j = 0; j < L; ++j)
                    if (s_ext[2 * j + 1] == k) acc += gam[2 * j + 1];
            }
            gr[t * ld_t + k] = gscale * (__expf(lg[t * ld_t + k] - s_lse[t]) - acc * ginv);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

int ns_for(int max_label_len) {
    const int S = 2 * max_label_len + 1;
    return S <= 64 ? 1 : (S <= 128 ? 2 : (S <= 256 ? 4 : 0));
}

size_t ctc_lds_bytes(int T, int max_label_len, int ns) {
    return ((size_t)T + 5 * 64 * ns + (size_t)T * (2 * max_label_len + 1)) * sizeof(float);
}

template <int NS>
int launch_ctc(const float* logits, int64_t ld_b, int64_t ld_t, int B, int T, int C, const int32_t* labels, int label_pitch,
               const int32_t* label_len, const int32_t* seq_len, int max_label_len, float gscale, float* loss, float* grad,
               float* ws, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)ctc_loss_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return AVSI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ctc_loss_kernel<NS>, dim3(B), dim3(CTPB), lds, st, logits, ld_b, ld_t, T, C, labels, label_pitch,
                       label_len, seq_len, max_label_len, gscale, loss, grad, ws);
    return avsi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Beam-search decoder (host).  Restates TensorFlow's CTCBeamSearchDecoder<>::Step / TopPaths
// (core/util/ctc/ctc_beam_search.h, TF 1.13-1.15) for top_paths = 1 with the default scorer.
struct BeamNode {
    int parent, label, first_child;
    float ob, ol, ot, nb, nl, nt;       // old / new probabilities: ending in blank, in a label, total (log)
};

const float LOG0 = -INFINITY;

float host_lse(float a, float b) {
    if (a == LOG0) return b;
    if (b == LOG0) return a;
    return a > b ? a + log1pf(expf(b - a)) : b + log1pf(expf(a - b));
}

void beam_search_one(const float* lg, int64_t ld_t, int n, int C, int width, bool merge, std::vector<int>& out,
                     float& score) {
    const int blank = C - 1;
    std::vector<BeamNode> nodes;
    nodes.push_back({-1, -1, -1, LOG0, LOG0, LOG0, 0.f, LOG0, 0.f});
    std::vector<int> leaves{0}, branches;
    std::vector<float> lp(C);
    for (int t = 0; t < n; ++t) {
        const float* x = lg + t * ld_t;
        float m = x[0];
        for (int k = 1; k < C; ++k) m = std::max(m, x[k]);
        float e = 0.f;
        for (int k = 0; k < C; ++k) e += expf(x[k] - m);
        const float off = m + logf(e);
        for (int k = 0; k < C; ++k) lp[k] = x[k] - off;

        branches = leaves;
        std::stable_sort(branches.begin(), branches.end(), [&](int p, int q) { return nodes[p].nt > nodes[q].nt; });
        leaves.clear();
        for (int bi : branches) {
            BeamNode& e2 = nodes[bi];
            e2.ob = e2.nb, e2.ol = e2.nl, e2.ot = e2.nt;
        }
        for (int bi : branches) {
            BeamNode& bn = nodes[bi];
            if (bn.parent >= 0) {
                const BeamNode& pa = nodes[bn.parent];
                if (pa.nt > LOG0) bn.nl = host_lse(bn.nl, bn.label == pa.label ? pa.ob : pa.ot);
                bn.nl += lp[bn.label];
            }
            bn.nb = bn.ot + lp[blank];
            bn.nt = host_lse(bn.nb, bn.nl);
            leaves.push_back(bi);
        }
        auto bottom = [&]() {
            int w = 0;
            for (int i = 1; i < (int)leaves.size(); ++i)
                if (nodes[leaves[i]].nt < nodes[leaves[w]].nt) w = i;
            return w;
        };
        auto is_candidate = [&](float total) {
            return total > LOG0 && ((int)leaves.size() < width || total > nodes[leaves[bottom()]].nt);
        };
        for (int bi : branches) {
            if (!is_candidate(nodes[bi].ot)) continue;
            if (nodes[bi].first_child < 0) {
                nodes[bi].first_child = (int)nodes.size();
                for (int k = 0; k < C - 1; ++k) nodes.push_back({bi, k, -1, LOG0, LOG0, LOG0, LOG0, LOG0, LOG0});
            }
            const BeamNode bn = nodes[bi];
            for (int k = 0; k < C - 1; ++k) {
                const int ci = bn.first_child + k;
                if (nodes[ci].nt > LOG0) continue;          // already in the beam
                const float prev = k == bn.label ? bn.ob : bn.ot;
                const float nl = lp[k] + prev;
                if (is_candidate(nl)) {
                    if ((int)leaves.size() == width) {
                        const int w = bottom();
                        BeamNode& dead = nodes[leaves[w]];
                        dead.nb = dead.nl = dead.nt = LOG0;
                        leaves.erase(leaves.begin() + w);
                    }
                    nodes[ci].nb = LOG0, nodes[ci].nl = nodes[ci].nt = nl;
                    leaves.push_back(ci);
                } else {
                    BeamNode& c = nodes[ci];
                    c.ob = c.ol = c.ot = c.nb = c.nl = c.nt = LOG0;
                }
            }
        }
    }
    int best = leaves[0];
    for (int li : leaves)
        if (nodes[li].nt > nodes[best].nt) best = li;
    score = nodes[best].nt;
    out.clear();
    for (int i = best; nodes[i].parent >= 0; i = nodes[i].parent) out.push_back(nodes[i].label);
    std::reverse(out.begin(), out.end());
    if (merge) out.erase(std::unique(out.begin(), out.end()), out.end());
}

}  // namespace

extern "C" size_t avsi_ctc_loss_workspace_bytes(int B, int T, int max_label_len) {
    const int ns = ns_for(max_label_len);
    if (B <= 0 || T <= 0 || max_label_len < 0 || !ns) return 0;
    return (size_t)B * 2 * T * 64 * ns * sizeof(float);
}

extern "C" int avsi_ctc_loss_f32(const float* logits, int64_t ld_b, int64_t ld_t, int B, int T, int C,
                                 const int32_t* labels, int label_pitch, const int32_t* label_len,
                                 const int32_t* seq_len, int max_label_len, float grad_scale, float* loss, float* grad,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!logits || !labels || !label_len || !seq_len || !loss || B <= 0 || T <= 0 || C < 2 || max_label_len < 0 ||
        label_pitch < max_label_len || ld_t < C || ld_b < (int64_t)T * ld_t)
        return AVSI_ERR_INVALID_ARG;
    const int ns = ns_for(max_label_len);
    const size_t lds = ns ? ctc_lds_bytes(T, max_label_len, ns) : 0;
    if (!ns || lds > 160 * 1024 - 256) return AVSI_ERR_UNSUPPORTED;     // labelling table must fit the CU's LDS
    if (!workspace || workspace_bytes < avsi_ctc_loss_workspace_bytes(B, T, max_label_len)) return AVSI_ERR_WORKSPACE;
    const hipStream_t st = (hipStream_t)stream;
    avsi_clear_error();
    float* ws = (float*)workspace;
    switch (ns) {
        case 1:
            return launch_ctc<1>(logits, ld_b, ld_t, B, T, C, labels, label_pitch, label_len, seq_len, max_label_len,
                                 grad_scale, loss, grad, ws, lds, st);
        case 2:
            return launch_ctc<2>(logits, ld_b, ld_t, B, T, C, labels, label_pitch, label_len, seq_len, max_label_len,
                                 grad_scale, loss, grad, ws, lds, st);
        default:
            return launch_ctc<4>(logits, ld_b, ld_t, B, T, C, labels, label_pitch, label_len, seq_len, max_label_len,
                                 grad_scale, loss, grad, ws, lds, st);
    }
}

extern "C" int avsi_ctc_beam_search_host_f32(const float* logits, int64_t ld_b, int64_t ld_t, int B, int T, int C,
                                             const int32_t* seq_len, int beam_width, int merge_repeated,
                                             int32_t* decoded, int decoded_pitch, int32_t* decoded_len, float* log_prob) {
    if (!logits || !seq_len || !decoded || !decoded_len || B <= 0 || T <= 0 || C < 2 || beam_width < 1 ||
        decoded_pitch < T || ld_t < C)
        return AVSI_ERR_INVALID_ARG;
    std::vector<int> out;
    for (int b = 0; b < B; ++b) {
        const int n = std::min(std::max(seq_len[b], 0), T);
        float score = 0.f;
        beam_search_one(logits + (int64_t)b * ld_b, ld_t, n, C, beam_width, merge_repeated != 0, out, score);
        for (int i = 0; i < decoded_pitch; ++i) decoded[(int64_t)b * decoded_pitch + i] = i < (int)out.size() ? out[i] : -1;
        decoded_len[b] = (int)out.size();
        if (log_prob) log_prob[b] = score;
    }
    return AVSI_OK;
}
