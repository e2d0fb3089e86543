/*
 * avsi_hip.h -- C ABI of the MI355X (gfx950) speech-inpainting hot path.
 *
 * The reference (dr-pato/audio-visual-speech-inpainting) has no FFI: its hot path is a
 * TensorFlow-1.x graph executed by sess.run().  Each entry point below replaces the TF
 * op(s) cited next to it (file:line relative to the reference's av_speech_inpainting/).
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 (AVSI_OK) or a negative avsi_status; nothing throws or aborts;
 *   - all tensor pointers are DEVICE pointers owned by the caller (e.g. torch data_ptr());
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it;
 *   - no hidden allocation, no global state: workspaces are caller-provided and sized by
 *     the matching *_workspace_bytes() query; distinct streams may be used concurrently;
 *   - float means IEEE binary32; "row" strides / leading dimensions are in ELEMENTS.
 */
#ifndef AVSI_HIP_H
#define AVSI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum avsi_status {
    AVSI_OK = 0,
    AVSI_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, bad stride           */
    AVSI_ERR_UNSUPPORTED = -2,   /* shape outside what the gfx950 kernels are built for   */
    AVSI_ERR_LAUNCH = -3,        /* hipGetLastError() != hipSuccess after a launch        */
    AVSI_ERR_WORKSPACE = -4      /* workspace null or too small                           */
} avsi_status;

/* ABI version (bumped on any signature change) and a static string for a status code. */
int avsi_abi_version(void);
const char* avsi_status_string(int status);
/* Host-only: AVSI_OK if the recurrent kernels take a stack of `num_layers` bidirectional LSTM layers of net_dim[l] units per
 * direction (config key net_dim; the reference takes any widths, models.py:95-99,107): any widths of 1 .. 256 units.
 * AVSI_ERR_UNSUPPORTED for more than 256 units. */
int avsi_blstm_net_supported(const int* net_dim, int num_layers);

/* ------------------------------------------------------------------------------------
 * Front end: framing + periodic-Hann window + rFFT + |.| + log + z-norm + mask (+ log-mel)
 * Replaces tf.contrib.signal.stft (audio_processing.py:35-36), get_spectrogram
 * (audio_processing.py:45-56), get_log_mel_spectrogram (audio_processing.py:59-72) and
 * the three elementwise ops of StackedBLSTMModel.__init__ (models.py:31-35), fused.
 * ------------------------------------------------------------------------------------ */

/* Number of floats of the per-(frame_len, nfft) constant table (window + FFT twiddles). */
size_t avsi_frontend_table_floats(int frame_len, int nfft);
/* Fill `table` (device, avsi_frontend_table_floats() floats) on `stream`. */
int avsi_frontend_init_tables(float* table, int frame_len, int nfft, void* stream);

typedef struct avsi_frontend_args {
    /* input waveform [B, N], row stride wav_stride (>= N) */
    const float* wav;
    int32_t batch;            /* B */
    int32_t num_samples;      /* N */
    int64_t wav_stride;
    /* framing: T = ceil(N / hop) frames exist; frames [0, num_frames) are produced */
    int32_t frame_len;        /* 384  (even, <= nfft) */
    int32_t hop;              /* 192  (even)          */
    int32_t nfft;             /* 512                  */
    int32_t num_frames;       /* <= ceil(N / hop): the out_shape[1] slice of get_stft */
    int32_t num_bins;         /* <= nfft/2+1: the out_shape[2] slice (audio_feat_dim) */
    const float* table;       /* from avsi_frontend_init_tables(frame_len, nfft) */
    /* optional per-bin normalisation (models.py:33); both null or both set */
    const float* mean;        /* [num_bins] */
    const float* stdev;       /* [num_bins] */
    /* optional mask [B, T, num_bins] (models.py:35) with element strides */
    const float* mask;
    int64_t mask_stride_b, mask_stride_t;
    /* outputs, each optional (null = skip); element strides per batch / per frame */
    float* out_stft;          /* interleaved re,im: [.., num_bins, 2]; strides count floats */
    int64_t stft_stride_b, stft_stride_t;
    float* out_spec;          /* |X|^power, log(. + eps) if log_spec, then (.-mean)/std */
    int64_t spec_stride_b, spec_stride_t;
    float* out_feat;          /* out_spec * mask */
    int64_t feat_stride_b, feat_stride_t;
    int32_t feat_cols;        /* >= num_bins: columns [num_bins, feat_cols) are zero-filled */
    float* out_logmel;        /* log(melW^T |X|^2 + eps) [.., num_mel] */
    int64_t logmel_stride_b, logmel_stride_t;
    /* band structure of the (sparse, triangular) mel matrix: band m has taps
       mel_w[m * mel_w_stride + j] on bins mel_start[m] + j, j < mel_len[m] */
    int32_t num_mel;
    const int32_t* mel_start;
    const int32_t* mel_len;
    const float* mel_w;
    int32_t mel_w_stride;
    /* spectrogram options (audio_processing.py:45-50) */
    float spec_power;         /* 1 = magnitude */
    int32_t log_spec;         /* 1 = log(. + eps) */
    float eps;                /* 1e-6 */
} avsi_frontend_args;

int avsi_frontend_f32(const avsi_frontend_args* args, void* stream);
/* With the inpainter's own shapes (nfft 512, 257 bins, mean / std, mask, log magnitude, 24 ms frames) and out_feat but NO
 * out_spec, the call stores the masked features only: 706 kB per utterance, exactly the algorithmic bytes. */

/* The L1 loss of the inpainter (reference models.py:144-151; as avsi_l1_loss_f32) with the TARGET recomputed from the waveform:
 * the front-end transform of `args` (same wav / table / mean / stdev / mask / geometry as the step's avsi_frontend_f32 call; its
 * out_* pointers are ignored) runs again and its epilogue compares with `pred` [batch][num_frames][257] (element strides given)
 * instead of storing the normalised target -- which the step then never has to write (257 kB per utterance) or read back:
 *   out3 = [mean |t - p|, sum |t - p| (1 - m) / sum (1 - m), sum |t - p| m / sum m];
 *   dpred (optional, pred's layout) = sign(p - t) * grad_scale.
 * workspace: avsi_l1_loss_workspace_bytes(n) bytes.  avsi_frontend_l1_loss_supported: 1 for the shapes it takes
 * (AVSI_ERR_UNSUPPORTED otherwise: use avsi_frontend_f32 with out_spec + avsi_l1_loss_f32). */
int avsi_frontend_l1_loss_supported(const avsi_frontend_args* args);
int avsi_frontend_l1_loss_f32(const avsi_frontend_args* args, const float* pred, int64_t pred_stride_b, int64_t pred_stride_t,
                              float* dpred, float grad_scale, float* out3, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32):
 *     C[M,N] = alpha * op(A) . op(B) + bias[n] + beta * C
 * transA = 0: A is [M,K] row-major (lda >= K);   transA = 1: A is [K,M] (lda >= M rounded to 4)
 * transB = 0: B is [K,N] row-major (ldb >= N rounded up to 4, padding addressable);
 * transB = 1: B is [N,K] (ldb >= K).  K, lda, ldb multiples of 4; A, B 16-byte aligned.
 * Replaces the time-batched tf.matmul pieces of the graph: the x_t half of LSTMBlockCell's
 * [x_t, h_{t-1}] . kernel product (models.py:107-115), the logits matmul (models.py:122), and
 * their gradients.  `row_map_*`: when row_map_bp > 0 the M rows are time-major (t, b) pairs with
 * batch pitch row_map_bp and output row (t, b) is stored at row b * row_map_t + t, rows with
 * b >= row_map_b are dropped (time-major -> reference [B, T, .] layout).
 * row_scale (optional, [M]) multiplies each finished row: tf.sequence_mask at models.py:136.
 * Two forms the call chooses by itself (results as documented; nothing for the caller to do):
 *   - A . B over many rows with N a multiple of 256 (the layer input projections): 128 x 256 output tiles; from 2048 tiles on
 *     the launch is the 512 RESIDENT workgroups, which take tile after tile from a queue of the library's own (a module global,
 *     zeroed on `stream` in front of the launch) -- not in a process that has declared a share of the chip (AVSI_COOP_CUS < 256);
 *   - N = 257 over many rows (the logits matmul onto the reference's 257 bins): the same tile for 256 columns, the last column
 *     taken by the same workgroups as a dot product over the rows they have staged (one launch instead of two).
 * ------------------------------------------------------------------------------------ */
typedef struct avsi_gemm_epilogue {
    const float* bias;        /* [N] or null */
    const float* row_scale;   /* [M] or null, applied after bias */
    int32_t row_map_bp, row_map_t, row_map_b;
    /* zero padding inside the reduction: the caller's promise that rows [k_zero[0], k_zero[1]) and [k_zero[2], k_zero[3])
     * of op(B) (or those columns of op(A)) are zero -- the 250 -> 256 padding of the two halves of a BLSTM layer's
     * input, the 257 -> 272 padding of the network input.  A . B on the 16-deep tiles then skips the MFMA steps that
     * multiply padding only (same result); all zeros = no promise; other forms ignore it */
    int32_t k_zero[4];
} avsi_gemm_epilogue;

int avsi_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                  const float* A, int64_t lda, const float* B, int64_t ldb,
                  float beta, float* C, int64_t ldc, const avsi_gemm_epilogue* epilogue,
                  void* stream);

/* ------------------------------------------------------------------------------------
 * Recurrent half of one bidirectional LSTM layer, all T steps, both directions.
 * Replaces the tf.while_loop over LSTMBlockCell / the CudnnLSTM op (models.py:95-115).
 *   xproj   [T][Bp][2][1024]  x_t . Wx + b for (fw, bw), packed gate columns
 *                             col = 128 w + 32 gate + u, hidden unit = 32 w + u, gate in (i,j,f,o)
 *   whp     [2][8][32][4][64][4]  recurrent kernel in MFMA fragment order (see blstm_fwd.hip)
 *   hout    [T][Bp][512]      fw hidden units at 0..255, bw at 256..511 (padded units are 0)
 *   reserve [T][Bp][2][5][256] or null: activated i, j, f, o and c_t kept for BPTT
 * Hidden size is padded to 256 (the reference uses 250); Bp must be a multiple of 32.
 * rows_per_wg: 32, 64, or 0 = choose from Bp.
 * ------------------------------------------------------------------------------------ */
int avsi_blstm_rec_fwd_f32(const float* xproj, const float* whp, float* hout, float* reserve,
                           int T, int Bp, int rows_per_wg, void* stream);
/* The same for the utterances [first_row, first_row + rows) of the batch only (multiples of 32; operands are still the
 * whole [T, Bp, ...] arrays; rows_per_wg 0 chooses from `rows`): the batch-stationary piece of a batch whose remainder
 * goes to a small-batch kernel (avsi_blstm_rec_fwd_coop_rows_f32 / avsi_blstm_rec_fwd_cs_rows_f32). */
int avsi_blstm_rec_fwd_rows_f32(const float* xproj, const float* whp, float* hout, float* reserve,
                                int T, int Bp, int rows_per_wg, int first_row, int rows, void* stream);

/* EXPLORATORY, not on the default path: C[M,N] = A[M,K] . B[K,N] + bias with every fp32 operand split into two bf16
 * values and the product taken as hi.hi + hi.lo + lo.hi on the bf16 matrix cores (fp32 accumulation).  Same role as
 * avsi_gemm_f32 for the layer input projections of models.py:95-115, selected by config['precision'] = 'bf16x3'.
 * B (the weights) is split and put into MFMA fragment order once, by avsi_pack_bf16x3_b, into a caller-owned buffer of
 * avsi_pack_bf16x3_b_bytes(K, N) bytes; A is split on the fly.  N % 128 == 0, K % 4 == 0, lda % 4 == 0, 16-byte aligned. */
size_t avsi_pack_bf16x3_b_bytes(int K, int N);
int avsi_pack_bf16x3_b(const float* B, int64_t ldb, int K, int N, void* packed, size_t packed_bytes, void* stream);
int avsi_gemm_bf16x3_f32(int M, int N, int K, const float* A, int64_t lda, const void* packed_b, const float* bias,
                         float* C, int64_t ldc, void* stream);

/* tf.nn.dropout(x, rate) on the last BLSTM layer's output in front of the projection (models.py:117): y = x * scale over
 * [rows][ld] with `cols` live columns, scale = 0 with probability rate, else 1 / (1 - rate), from a counter-based
 * generator (seed + element index).  `scale` is written out for the backward pass (avsi_scale_elements_f32: x *= scale). */
int avsi_dropout_f32(const float* x, float* y, float* scale, int64_t rows, int cols, int ld, float rate,
                     unsigned long long seed, void* stream);
int avsi_scale_elements_f32(float* x, const float* scale, int64_t rows, int cols, int ld, void* stream);

/* ------------------------------------------------------------------------------------
 * L1 loss and its diagnostics over [n] elements (models.py:144-151):
 *   out[0] = mean|t-p|, out[1] = sum|t-p|(1-m)/sum(1-m), out[2] = sum|t-p|m/sum(m)
 * and, when dpred != null, dpred = grad_scale * sign(p - t) (the tf.abs gradient).
 * workspace: avsi_l1_loss_workspace_bytes(n) bytes.
 * ------------------------------------------------------------------------------------ */
size_t avsi_l1_loss_workspace_bytes(int64_t n);
int avsi_l1_loss_f32(const float* target, const float* pred, const float* mask, int64_t n,
                     float* out3, float* dpred, float grad_scale, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Small-batch form of avsi_blstm_rec_fwd_f32 (same operands and results): every (32-utterance
 * tile, direction) pair is spread over `split` = 4, 8, 16 or 32 workgroups that keep their piece of the
 * recurrent kernel in registers for all T steps and exchange h_t through hout with a per-step
 * counter in `workspace` (avsi_blstm_rec_fwd_coop_workspace_bytes(Bp) bytes, see below).
 * A launch must be wholly resident (one workgroup per CU), so batches beyond 128 (split 32) / 256 (16) /
 * 512 (8) / 1024 (4) utterances run as consecutive launches over tile ranges.  `max_cus` (<= 0: all 256) is the
 * number of compute units the caller grants one launch: a process that keeps other kernels in flight beside the
 * recurrence (RCCL collectives of data-parallel training, batches on other streams) passes what is left, and the
 * launches are cut to fit; AVSI_ERR_UNSUPPORTED when not even one tile (2 * split workgroups) fits.
 * `split` = 64: the 32-way form on two independent 16-row halves per tile (64 workgroups per (tile, direction): two tiles per
 * launch on the whole chip; v_mfma_f32_16x16x4_f32, half the matrix work per workgroup and step).  It exchanges through the
 * exchange copy only: AVSI_ERR_WORKSPACE without the optional part of the workspace described below.
 * The workspace is zeroed ONCE by the caller, when it is allocated: word 0 is a STICKY status that no call clears,
 * the counters behind it (one 256-byte line each) are put back to zero by the kernels themselves when a launch ends,
 * so consecutive calls on one stream need nothing in between.  Once the stream has drained, a non-zero word 0 means
 * some launch since the allocation had a workgroup stop waiting for its peers: its outputs are invalid and it may
 * have left counters behind -- zero the whole workspace before using it again.  Every wait is bounded in WALL-CLOCK
 * time (AVSI_COOP_TIMEOUT_MS, default 2000: a group whose members cannot all be resident -- another kernel holds the
 * LDS of a CU it needs; the 16- and 32-way splits keep a group on ONE XCD and need its CUs to themselves -- gives up
 * after that long), a waiting workgroup also leaves as soon as any other has given up, and a launch that finds word 0
 * already set returns at once without touching its outputs: one conflict costs one bound, whatever is enqueued
 * behind it.  avsi_step_guard_f32 / avsi_adam_tf_guarded_f32 keep such a step away from the variables. */
size_t avsi_blstm_rec_fwd_coop_workspace_bytes(int Bp);
/* Optional: a workspace of AVSI_COOP_EXCHANGE_OFFSET + avsi_blstm_rec_fwd_coop_exchange_bytes(T, Bp) bytes lets the
 * forward call with split 16 / 32 exchange h through a copy in a layout of its own (whole-line stores, contiguous
 * fragment loads: 0.86 -> 0.78 ms per layer at 32 utterances on buffers no cache holds).  The copy lives at that FIXED
 * offset, clear of the counters of any batch size, and needs no initialisation.  Results are identical.
 * (AVSI_COOP_NOACK=1 in the environment, opt-in: the call presets the copy to an all-ones pattern itself and the members
 * publish without waiting for their store's acknowledgement -- a reader re-loads words that still hold the pattern.) */
#define AVSI_COOP_EXCHANGE_OFFSET ((size_t)1 << 20)
size_t avsi_blstm_rec_fwd_coop_exchange_bytes(int T, int Bp);
/* The same for avsi_blstm_rec_bwd_coop_f32 with split 16 / 32 (dz in exchange layout; same offset, the two calls may
 * share one workspace: a launch is done with its copy when it ends). */
size_t avsi_blstm_rec_bwd_coop_exchange_bytes(int T, int Bp);
int avsi_blstm_rec_fwd_coop_f32(const float* xproj, const float* whp, float* hout, float* reserve,
                                int T, int Bp, int split, int max_cus, void* workspace, size_t workspace_bytes,
                                void* stream);
/* The same for the utterances [first_row, first_row + rows) of the batch only (both multiples of 32; the operands are
 * still the whole [T, Bp, ...] arrays): lets a caller give the remainder of a batch to the kernel of ITS size instead of a
 * second launch of the large one (cooperative launches are latency-bound: 250 steps whatever the number of groups). */
int avsi_blstm_rec_fwd_coop_rows_f32(const float* xproj, const float* whp, float* hout, float* reserve,
                                     int T, int Bp, int split, int first_row, int rows, int max_cus, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* Small-batch form of avsi_blstm_rec_bwd_f32 (same operands and results), the gradient of the
 * cooperative forward above: same group / split / workspace / residency rules (split 32 here = 16 slices of
 * the hidden state x 2 halves of the tile's 32 utterances), dz doubles as the exchange buffer between the
 * workgroups of a group. */
int avsi_blstm_rec_bwd_coop_f32(const float* dhout, const float* reserve, const float* whbT, float* dz,
                                int T, int Bp, int split, int max_cus, void* workspace, size_t workspace_bytes,
                                void* stream);

/* Mid-batch form of avsi_blstm_rec_fwd_f32 (same operands and results; reference models.py:95-115): groups of 8
 * workgroups own (`rows_per_group` = 16 or 32 utterances, direction); the waves of a workgroup split the gate COLUMNS
 * (4 hidden units each, their piece of the recurrent kernel in registers for all T steps), so there are no partial
 * sums to park and two workgroups share a compute unit -- one's wait for its peers hides under the other's MFMAs.
 * Same exchange protocol, workspace word 0 and `max_cus` meaning as avsi_blstm_rec_fwd_coop_f32; a launch holds
 * 2 * max_cus workgroups (the occupancy the runtime reports), larger batches run as consecutive launches. */
size_t avsi_blstm_rec_fwd_cs_workspace_bytes(int Bp);
/* (utterance tile, direction) groups one launch of the kernel holds on `max_cus` compute units (needs a GPU). */
int avsi_blstm_rec_fwd_cs_groups_per_launch(int rows_per_group, int with_reserve, int max_cus);
int avsi_blstm_rec_fwd_cs_f32(const float* xproj, const float* whp, float* hout, float* reserve,
                              int T, int Bp, int rows_per_group, int max_cus, void* workspace,
                              size_t workspace_bytes, void* stream);
/* The same for the utterances [first_row, first_row + rows) only (multiples of rows_per_group); see
 * avsi_blstm_rec_fwd_coop_rows_f32. */
int avsi_blstm_rec_fwd_cs_rows_f32(const float* xproj, const float* whp, float* hout, float* reserve,
                                   int T, int Bp, int rows_per_group, int first_row, int rows, int max_cus,
                                   void* workspace, size_t workspace_bytes, void* stream);

/* Diagnostic: while `buffer` (device memory, 32 * 8 * 2 * 8 uint64) is set, avsi_blstm_rec_fwd_cs_f32 and the 32-way
 * cooperative forward kernel without reserve (avsi_blstm_rec_fwd_coop_f32, split 32) record the
 * 100 MHz wall clock at eight phases of steps 64 .. 71 for waves 0 and 1 of their first 32 workgroups; NULL ends it. */
int avsi_diag_cs_stamps(void* buffer);

/* Holds `stream` back for `microseconds` (0 .. 1000) with one idle wave: lets a kernel on another stream that becomes
 * ready at the same instant (a cooperative recurrent grid, which must be wholly resident) be dispatched first. */
int avsi_stream_delay_us(int microseconds, void* stream);

/* Diagnostic: park `num_cus` workgroups, each claiming a whole compute unit (160 KiB of LDS), on `stream` until
 * *release (device int32) becomes non-zero or ~`max_ms` milliseconds have passed (every workgroup leaves by itself).
 * Stands in for the CUs an RCCL collective holds while the cooperative kernels run (tests/test_coop_residency_gpu.py). */
int avsi_diag_occupy_cus(int num_cus, const int* release, int max_ms, void* stream);
/* dst[0 .. n) = src[0 .. n) with 16-byte accesses and a chip-sized grid: the streaming-copy yardstick bench.py quotes
 * the HBM-bound kernels against (`device_copy_GB/s`).  n a multiple of 4, both pointers 16-byte aligned. */
int avsi_diag_copy_f32(const float* src, float* dst, int64_t n, void* stream);

/* Loss of the speaker-embedding model variants (reference models.py:1006-1029 StackedBLSTMSSNNModel,
 * :1367-1394 StackedBLSTMEmbeddingModel): the prediction keeps the known bins,
 *   prediction = seq_mask * (target * mask + logits * (1 - mask)),
 * and the training loss is loss_hole.  pred_inout [n] holds seq_mask * logits on entry (the
 * projection GEMM's row-scale epilogue) and the prediction on return; row_scale [n / row_len] is the
 * sequence mask per (utterance, frame) row of row_len bins, or null for all ones.
 *   out4[0..2] as avsi_l1_loss_f32 (taken on the prediction), out4[3] = 1 / sum(1 - mask);
 *   dlogits (optional) [n] = sign(prediction - target) (1 - mask)^2 / sum(1 - mask)
 *   = d loss_hole / d logits up to the sequence mask the caller folds in afterwards.
 * workspace: avsi_l1_loss_workspace_bytes(n) bytes. */
int avsi_l1_loss_blend_f32(const float* target, float* pred_inout, const float* mask,
                           const float* row_scale, int row_len, int64_t n, float* out4,
                           float* dlogits, void* workspace, size_t workspace_bytes, void* stream);

/* CTC head of the multi-task models (reference models.py:1944-1964 StackedBLSTMSSNNCTCLossModel.loss,
 * :1634-1654 StackedBLSTMCTCLossModel.loss): tf.nn.ctc_loss(labels, logits, sequence_length,
 * preprocess_collapse_repeated=False, ctc_merge_repeated=True) and its gradient.
 *   logits [B][T][C] un-normalised (softmax is applied inside, like TF), element (b, t, k) at
 *          b*ld_b + t*ld_t + k; the blank label is C - 1;
 *   labels int32 [B][label_pitch], row b's first label_len[b] entries (the dense form the reference
 *          feeds, ctc_label_dense_to_sparse :1760); max_label_len >= every label_len[b] (<= 127);
 *   seq_len int32 [B]: frames t >= seq_len[b] are ignored and get zero gradient;
 *   loss [B] = -log p(labels_b | logits_b); +inf (and a zero gradient row) when the labelling does not
 *          fit into seq_len[b] frames (TF raises instead -- the trainers stop on an infinite loss);
 *   grad (optional, same strides as logits) = grad_scale * d loss[b] / d logits[b]
 *          = grad_scale * (softmax - posterior of the labelling's states per class).
 * Out-of-range labels are clamped into [0, C-1] (the host mirror rejects them before the call).
 * workspace: avsi_ctc_loss_workspace_bytes (0 = unsupported sizes) -- the alpha / beta tables.
 * AVSI_ERR_UNSUPPORTED if T * (2 max_label_len + 1) floats do not fit the CU's LDS. */
size_t avsi_ctc_loss_workspace_bytes(int B, int T, int max_label_len);
int avsi_ctc_loss_f32(const float* logits, int64_t ld_b, int64_t ld_t, int B, int T, int C,
                      const int32_t* labels, int label_pitch, const int32_t* label_len,
                      const int32_t* seq_len, int max_label_len, float grad_scale, float* loss, float* grad,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Host helper (no GPU work; HOST pointers): tf.nn.ctc_beam_search_decoder(logits, seq_len, beam_width,
 * top_paths=1, merge_repeated) of the reference's `decoding` / `per` diagnostics (models.py:1934-1942,
 * 2026-2031) -- TensorFlow runs this decoder on the CPU too.  decoded int32 [B][decoded_pitch >= T]
 * padded with -1 (tf.sparse.to_dense(default_value=-1)), decoded_len [B], log_prob [B] (optional) =
 * log probability of the returned labelling under the beam. */
int avsi_ctc_beam_search_host_f32(const float* logits, int64_t ld_b, int64_t ld_t, int B, int T, int C,
                                  const int32_t* seq_len, int beam_width, int merge_repeated,
                                  int32_t* decoded, int decoded_pitch, int32_t* decoded_len, float* log_prob);

/* Split-K form of avsi_gemm_f32 for reductions over very many rows (weight gradients
 * dW = X^T . dZ over all T*Bp rows): K is cut into `splits` chunks, partial [M,N] slabs go to
 * `workspace` (avsi_gemm_splitk_workspace_bytes), then are summed in chunk order (deterministic,
 * no atomics) into C [M,N] (contiguous, ldc = N).  No bias / epilogue options. */
size_t avsi_gemm_splitk_workspace_bytes(int M, int N, int splits);
int avsi_gemm_splitk_f32(int transA, int transB, int M, int N, int K, float alpha,
                         const float* A, int64_t lda, const float* B, int64_t ldb,
                         float* C, int splits, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * BPTT through the recurrent half of one bidirectional LSTM layer (gradient of
 * avsi_blstm_rec_fwd_f32; the reference gets it from tf.gradients over the while_loop /
 * the CudnnLSTM backward op, models.py:95-115,178).
 *   dhout   [T][Bp][512]        d loss / d layer output
 *   reserve [T][Bp][2][5][256]  from the forward call
 *   whbT    [2][8][128][64][4]  recurrent kernel, transposed-product fragment order
 *   dz      [T][Bp][2][1024]    OUT: d loss / d gate pre-activations, packed gate columns
 * The time-batched gradients follow as GEMMs / column sums over dz.
 * ------------------------------------------------------------------------------------ */
int avsi_blstm_rec_bwd_f32(const float* dhout, const float* reserve, const float* whbT, float* dz,
                           int T, int Bp, void* stream);
/* Host-only: the name of the kernel avsi_blstm_rec_bwd_f32 launches for a padded batch of Bp utterances under the current
 * environment (AVSI_BWD_PP, AVSI_BWD_KH) -- "blstm_rec_bwd_pp_kernel", "blstm_rec_bwd_kh_kernel" or "blstm_rec_bwd_kernel" --
 * so that a profile or a bench line names what actually ran.  A static string, never null. */
const char* avsi_blstm_rec_bwd_kernel_name(int Bp);

/* ------------------------------------------------------------------------------------
 * Memory-bound helpers.
 * avsi_relayout_rows_f32: dst[b*dsb + t*dst + c] = (c < C ? src[b*ssb + t*sst + c] * scale(b,t) : 0)
 *   for c < dst_cols: moves [B,T,C] feed tensors into the padded time-major layout (and
 *   gradients back), folding tf.sequence_mask (models.py:136) in as the row scale.
 * avsi_colsum_f32: out[n] = sum_m x[m*ld + n] (bias gradients), deterministic two-stage.
 * avsi_adam_tf_f32: tf.train.AdamOptimizer(lr, beta1, beta2, eps) step `step` (1-based) on flat
 *   buffers (models.py:168): g' = grad*grad_scale + l2*p; m,v update; p -= lr_t m / (sqrt(v)+eps).
 * ------------------------------------------------------------------------------------ */
int avsi_relayout_rows_f32(const float* src, int64_t src_stride_b, int64_t src_stride_t,
                           float* dst, int64_t dst_stride_b, int64_t dst_stride_t,
                           int B, int T, int C, int dst_cols,
                           const float* row_scale, int64_t scale_stride_b, int64_t scale_stride_t,
                           void* stream);
size_t avsi_colsum_workspace_bytes(int64_t M, int N);
int avsi_colsum_f32(const float* x, int64_t ld, int64_t M, int N, float* out,
                    void* workspace, size_t workspace_bytes, void* stream);
int avsi_adam_tf_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                     float lr, float beta1, float beta2, float eps, int64_t step,
                     float grad_scale, float l2, void* stream);
/* Step guard (no reference counterpart: the reference's only failure rule is the host-side NaN / Inf abort AFTER the
 * update, training.py:244-249; here the variables are never touched by a void gradient).
 * avsi_step_guard_f32: out2[0] = 0 if *loss is finite (or loss is NULL) else NaN; out2[1] = 1 if *status_a or
 *   *status_b (device int32 words, either may be NULL: the sticky status words of the cooperative recurrent
 *   launches) is non-zero, else 0.  Data-parallel ranks carry the two words through the last gradient all-reduce
 *   bucket (sum), so every rank sees the same verdict.
 * avsi_adam_tf_guarded_f32: avsi_adam_tf_f32 that reads `n_skip` (<= 8) device floats first and does NOTHING when any
 *   of them is not exactly 0 (NaN counts): param, m, v stay as they were. */
int avsi_step_guard_f32(const float* loss, const int* status_a, const int* status_b, float* out2, void* stream);
int avsi_adam_tf_guarded_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                             float lr, float beta1, float beta2, float eps, int64_t step,
                             float grad_scale, float l2, const float* skip, int n_skip, void* stream);
/* avsi_sgd_momentum_f32: the reference's other two optimizer_type choices behind the same guard (models.py:170-176,
 *   tf.train.GradientDescentOptimizer(lr) / tf.train.MomentumOptimizer(lr, momentum = 0.9)): g' = grad * grad_scale + l2 * p;
 *   accum == NULL: p -= lr * g' (sgd); otherwise TF's momentum form accum = momentum * accum + g', p -= lr * accum (the
 *   rate multiplies the accumulator when it is applied; no Nesterov term).  `lr` is the staircase-decayed rate of the step
 *   (tf.train.exponential_decay, models.py:165-166), computed by the caller.  `skip` as for avsi_adam_tf_guarded_f32. */
int avsi_sgd_momentum_f32(float* param, const float* grad, float* accum, int64_t n, float lr, float momentum,
                          float grad_scale, float l2, const float* skip, int n_skip, void* stream);

/* ------------------------------------------------------------------------------------
 * Inverse STFT / waveform reconstruction: tf.contrib.signal.inverse_stft with
 * inverse_stft_window_fn(hop) (audio_processing.py:145-157), optionally fused with the polar
 * assembly of get_sources (audio_processing.py:160-164) and with the de-normalisation + masked
 * phase of StackedBLSTMModel.enhanced_sources (models.py:181-197).
 *   mode 0: in0 = complex spectrogram [B][T][num_bins][2] (strides count floats)
 *   mode 1: in0 = magnitude, in1 = phase, both [B][T][num_bins] with the in0 strides
 *   mode 2: in0 = prediction (normalised log-magnitude), mean/stdev optional de-normalisation,
 *           in1 = target STFT (complex, its own strides), in2 = mask [B][T][num_bins] or null
 *           (null = oracle phase): X = exp(in0*std+mean) * (S m)/|S m|, angle(0) = 0.
 *   mode 3: mode 2 with the target WAVEFORM in place of its STFT: `wav` [B][wav_samples] (row stride wav_stride_b);
 *           S = STFT(wav) with the front end's framing (periodic Hann of frame_len, hop, zero-padded tail,
 *           audio_processing.py:25-42) is computed inside the kernel, frame tile by frame tile, and never stored:
 *           StackedBLSTMModel.enhanced_sources (models.py:181-197) for 0.9 MB per utterance instead of 2.9 MB.
 * Output [B][num_samples] (row stride out_stride_b), num_samples <= (T-1) hop + frame_len.
 * Requires nfft = 512 and hop <= frame_len <= 2 hop.
 * ------------------------------------------------------------------------------------ */
typedef struct avsi_istft_args {
    int32_t mode;
    const float* in0;
    int64_t in_stride_b, in_stride_t;
    const float* in1;
    int64_t in1_stride_b, in1_stride_t;
    const float* in2;
    int64_t in2_stride_b, in2_stride_t;
    const float* mean;
    const float* stdev;
    int32_t batch, num_frames, num_bins, frame_len, hop, nfft;
    const float* table;       /* from avsi_istft_init_tables(frame_len, hop, nfft) */
    float* out;
    int64_t out_stride_b;
    int64_t num_samples;
    const float* wav;         /* mode 3 only */
    int64_t wav_stride_b, wav_samples;
} avsi_istft_args;

size_t avsi_istft_table_floats(int frame_len, int hop, int nfft);
int avsi_istft_init_tables(float* table, int frame_len, int hop, int nfft, void* stream);
int avsi_istft_f32(const avsi_istft_args* args, void* stream);

/* ------------------------------------------------------------------------------------
 * LWS ("local weighted sums") phase reconstruction: the refinement `infer` applies to every enhanced waveform
 * when --oracle_phase is not given (inference.py:119,141-154), there through the third-party `lws` package
 * (lws.lws(384, 192, fftsize=512, mode='speech'): .stft / .run_lws / .istft).  The package is not part of the
 * reference tree; these entry points implement the published algorithm (Le Roux et al., DAFx-10 / ASJ 2010) under
 * the conventions written down in oracle/lws.py.  UNPINNED against the package.
 *   spectrograms are [B][num_frames][nfft/2+1][2] floats (complex, contiguous), num_frames =
 *   avsi_lws_num_frames(num_samples, hop, nfft) ('perfectrec' padding: nfft - hop zeros either side);
 *   windows: sqrt of the symmetric Hann window of frame_len samples, zero-padded symmetrically to nfft,
 *   and its perfect-reconstruction synthesis window (table from avsi_lws_init_tables).
 *   Supported geometry: nfft = 512, hop <= frame_len <= 2 hop, 64 hop / nfft integer.
 *   avsi_lws_stft_f32    lws.stft   (inference.py:143)
 *   avsi_lws_stitch_f32  ref = null: S <- |S| exp(j angle(S) mask_adj)                       (inference.py:144-147)
 *                        ref given:  S <- |S| exp(j (angle(ref) + angle(S) (1 - mask_adj)))  (inference.py:149-152)
 *                        mask_adj = mask [B][mask_frames][mask_bins] (element strides), zero outside it
 *   avsi_lws_run_f32     lws.run_lws (inference.py:148): nofuture / online / batch sweeps in place; thresholds
 *                        alpha exp(-beta j^gamma) relative to the mean magnitude of each utterance;
 *                        utterances_per_wave (1, 2, 4) x waves_per_group (1, 4, 8, 16; > 1 only with one utterance per
 *                        wave) x groups_per_utterance (workgroups, each a CU at 16 waves): the sweeps of an utterance
 *                        run as a pipeline over that many waves -- same results; 0 = chosen from the batch.
 *                        workspace (avsi_lws_run_workspace_bytes(batch), zeroed by the call): word 0 is the status --
 *                        non-zero after the stream has drained = a pipeline stage stopped waiting for its
 *                        predecessor (all workgroups of an utterance must be resident together), outputs invalid
 *   avsi_lws_run_skew_f32  the same sweeps (same arguments, same results up to the order of the 33 taps' sum) with the
 *                        FRAMES of a sweep in the lanes of a wave, six bins apart (csrc/lws_skew.hip): the default of the
 *                        host layer.  waves_per_group (4, 8, 16) consecutive sweeps of ONE utterance per
 *                        workgroup, groups_per_utterance workgroups chained per utterance; 0 = the cheapest shape under
 *                        the library's pipeline model (avsi_lws_skew_launch_shape tells which: a host-side query, no
 *                        GPU work); results do not depend on either.  The workspace
 *                        (avsi_lws_run_skew_workspace_bytes(batch, num_frames), ~1.4 MB per utterance of 252 frames)
 *                        holds the spectrogram in the kernel's diagonal layout; word 0 is the status as above
 *   avsi_lws_run_duo_f32   the same sweeps with TWO utterances in the lanes of a wave, 32 frames each, nine bins apart, the
 *                        neighbour rows' sums handed on through LDS and memory (csrc/lws_duo.hip): every lane busy, against
 *                        45 of 64 in the skewed kernel -- the large-batch form (the host layer takes it from
 *                        AVSI_LWS_DUO_MIN utterances), bit-identical results.  The reference's geometry only
 *                        (384 / 192 / 512, L = 5; others: AVSI_ERR_UNSUPPORTED, take avsi_lws_run_skew_f32).
 *                        waves_per_group (4, 8, 16; 0 = 16) consecutive sweeps per workgroup, groups_per_pair workgroups
 *                        chained per two utterances (0 = as many as fill the chip); workspace
 *                        avsi_lws_run_duo_workspace_bytes(batch, num_frames), ~1.0 MB per utterance of 252 frames
 *   avsi_lws_istft_f32   lws.istft  (inference.py:153): out [B][out_samples], out_samples <=
 *                        (num_frames - 1) hop + nfft - 2 (nfft - hop); workspace from avsi_lws_istft_workspace_bytes
 * ------------------------------------------------------------------------------------ */
int avsi_lws_num_frames(int num_samples, int hop, int nfft);
size_t avsi_lws_table_floats(int frame_len, int hop, int nfft);
int avsi_lws_init_tables(float* table, int frame_len, int hop, int nfft, void* stream);
int avsi_lws_stft_f32(const float* wav, int64_t wav_stride, int batch, int num_samples, const float* table, int hop,
                      int nfft, float* spec, int num_frames, void* stream);
int avsi_lws_stitch_f32(float* spec, const float* ref, const float* mask, int64_t mask_stride_b, int64_t mask_stride_t,
                        int mask_frames, int mask_bins, int batch, int num_frames, int nfft, void* stream);
int avsi_lws_run_f32(float* spec, int batch, int num_frames, int frame_len, int hop, int nfft, int L,
                     int nofuture_iterations, float nofuture_alpha, int online_iterations, float online_alpha,
                     int batch_iterations, float batch_alpha, float batch_beta, float batch_gamma,
                     int utterances_per_wave, int waves_per_group, int groups_per_utterance, void* workspace,
                     size_t workspace_bytes, void* stream);
size_t avsi_lws_run_workspace_bytes(int batch);
int avsi_lws_run_skew_f32(float* spec, int batch, int num_frames, int frame_len, int hop, int nfft, int L,
                          int nofuture_iterations, float nofuture_alpha, int online_iterations, float online_alpha,
                          int batch_iterations, float batch_alpha, float batch_beta, float batch_gamma,
                          int waves_per_group, int groups_per_utterance, void* workspace, size_t workspace_bytes,
                          void* stream);
size_t avsi_lws_run_skew_workspace_bytes(int batch, int num_frames);
int avsi_lws_skew_launch_shape(int batch, int num_frames, int sweeps, int* waves_per_group, int* groups_per_utterance);
int avsi_lws_run_duo_f32(float* spec, int batch, int num_frames, int frame_len, int hop, int nfft, int L,
                          int nofuture_iterations, float nofuture_alpha, int online_iterations, float online_alpha,
                          int batch_iterations, float batch_alpha, float batch_beta, float batch_gamma,
                          int waves_per_group, int groups_per_pair, void* workspace, size_t workspace_bytes, void* stream);
size_t avsi_lws_run_duo_workspace_bytes(int batch, int num_frames);
size_t avsi_lws_istft_workspace_bytes(int batch, int num_frames, int nfft);
int avsi_lws_istft_f32(const float* spec, int batch, int num_frames, const float* table, int hop, int nfft, float* out,
                       int64_t out_stride, int out_samples, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * U-Net building blocks (models.py:519-715 UNetFConvModel, unet_layers.py:6-37).  Activations are
 * NHWC as 2-D [B*H*W][C] with row pitch ld.  A SAME / stride-1 tf.nn.conv2d is avsi_im2col_f32 +
 * avsi_gemm_f32 with the TF filter [kh][kw][Cin][Cout] as the B matrix (rows padded to Kc, a
 * multiple of 4); its gradients are avsi_gemm_splitk_f32 (filter) and avsi_gemm_f32 +
 * avsi_col2im_f32 (input).  im2col can read a second, half-resolution source through a nearest
 * 2x up-sampling and appends its channels after src0's (UpSampling2D + concat, unet_layers.py:28-29).
 * Batch normalisation uses batch statistics (training=True, biased variance, eps = 1e-3);
 * act: 0 none, 1 relu, 2 leaky_relu(0.2), 3 leaky_relu(0.3) (the speaker-embedding MLP,
 * models.py:822-824).  Pass mean = null for layers without batch norm.
 * workspace for the reductions: avsi_unet_workspace_bytes(C).
 * ------------------------------------------------------------------------------------ */
/* Convolution as an implicit GEMM (no im2col matrix): out [B*H*W][ldo] = conv2d(concat(src0,
 * up2x(src1_coarse)), filter [k*k*(C0+C1)][ldf]) + bias, SAME padding, stride 1.  The operand rows
 * are gathered by the GEMM kernel's LDS-DMA loads straight from the NHWC activations; taps outside
 * the image read `zeros64` (>= 64 bytes of zeros, 16-byte aligned, caller-owned).  Needs C0 and C1
 * to be multiples of 16 (every 16-deep reduction tile then lies in one tap of one source) --
 * AVSI_ERR_UNSUPPORTED otherwise: use avsi_im2col_f32 + avsi_gemm_f32 for those layers. */
int avsi_conv2d_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1,
                    int B, int H, int W, int k, const float* filter, int ldf, const float* bias,
                    int Cout, float* out, int ldo, const float* zeros64, void* stream);
/* The same with the reduction over (tap, channel) cut into `splits` chunks whose partial slabs (workspace) are summed in chunk
 * order by a second kernel: for layers whose output is only a few 128-row tiles (the 128-channel U-Net layers at the
 * reference's batch of 32: 4 .. 64 tiles on 256 CUs).  avsi_conv2d_splitk_suggest gives the number of chunks that fills the chip
 * about once with at least eight 16-deep k-tiles per workgroup (1 = do not split); splits < 2, ldo != Cout or a workspace
 * smaller than avsi_conv2d_splitk_workspace_bytes: the plain launch.  Same results up to the order of the sum. */
size_t avsi_conv2d_splitk_workspace_bytes(int B, int H, int W, int ldo, int splits);
int avsi_conv2d_splitk_suggest(int B, int H, int W, int k, int C0, int C1, int Cout);
int avsi_conv2d_splitk_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H,
                           int W, int k, const float* filter, int ldf, const float* bias, int Cout, float* out, int ldo,
                           const float* zeros64, int splits, void* workspace, size_t workspace_bytes, void* stream);
/* Inference tail of the U-Net in one call: d6 = conv3x3(concat(src0 [1 channel], up2x(src1 [16 channels])), 17 -> 1) with the
 * batch statistics of its one channel taken in the convolution's epilogue, then one pass: batch norm (gamma, beta, eps) +
 * LeakyReLU(0.2) (unet_layers.py:33-35) + the 1 x 1 output convolution (w_out[0], b_out[0]: models.py:607) + the sequence
 * mask of `prediction` (seq_len int64 [B], frames >= seq_len[b] zeroed: models.py:609-615).  src1_bn as in
 * avsi_conv2d_bn_f32 (src1_coarse = the raw 16-channel output of the layer below).  conv [B*H*W] is scratch,
 * logits [B*H*W] (the un-masked `inference`) may be NULL, pred [B*H*W].  H % 8 == 0, W % 32 == 0. */
size_t avsi_unet_tail_workspace_bytes(int B, int H, int W);
int avsi_unet_tail_f32(const float* src0, int ld0, const float* src1_coarse, int ld1, const float* const* src1_bn, int B, int H,
                       int W, const float* filter, int ldf, const float* bias, const float* gamma, const float* beta, float eps,
                       const float* w_out, const float* b_out, const long long* seq_len, float* conv, float* logits,
                       float* pred, void* workspace, size_t workspace_bytes, void* stream);
/* Direct form for the thin full-resolution layers no MFMA shape fits: (k, C0, C1, Cout) in
 * {(7, 1, 0, 16), (3, 1, 16, 1), (1, 1, 0, 1)} (unet_layers.py / models.py:592,605,607);
 * AVSI_ERR_UNSUPPORTED for anything else. */
int avsi_conv2d_thin_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1,
                         int B, int H, int W, int k, const float* filter, int ldf, const float* bias,
                         int Cout, float* out, int ldo, void* stream);
/* Second half of the input gradient of a convolution computed as an implicit GEMM (dX of
 * concat(src0, up2x(src1)) = avsi_conv2d_f32 of dY with the tap-flipped, transposed filter, one
 * [B*H*W][C0 + C1] matrix): channels [0, C0) are copied / added to dsrc0, channels [C0, C0 + C1)
 * are summed over every 2x2 block into dsrc1_coarse (adjoint of the nearest up-sampling).  Either
 * target may be null.  Channel counts and pitches must be multiples of 4. */
int avsi_split_sumpool_f32(const float* dx, int ldx, float* dsrc0, int C0, int ld0, int accumulate0,
                           float* dsrc1_coarse, int C1, int ld1, int accumulate1, int B, int H, int W,
                           void* stream);
/* Filter gradient of the same convolution, again without the im2col matrix:
 *   dw [k*k*(C0+C1)][ldw] = im2col(concat(src0, up2x(src1)))^T . dy [B*H*W][ldy]
 * (ldw = Cout rounded up to 4).  The reduction over the pixels is cut into `splits` slabs in
 * `workspace` (avsi_conv2d_wgrad_workspace_bytes) and summed in order.  Needs C0, C1 multiples of 4
 * and B*H*W a multiple of 16; AVSI_ERR_UNSUPPORTED otherwise. */
size_t avsi_conv2d_wgrad_workspace_bytes(int C0, int C1, int k, int Cout, int splits);
int avsi_conv2d_wgrad_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1,
                          int B, int H, int W, int k, const float* dy, int ldy, int Cout, float* dw, int ldw,
                          int splits, const float* zeros64, void* workspace, size_t workspace_bytes,
                          void* stream);
/* Backward of the thin layers (same (k, C0, C1, Cout) set as avsi_conv2d_thin_f32), direct form:
 * filter gradient dw [k*k*(C0+C1)][ldw] (per-block partials in `workspace`, summed in order), and
 * for the 1 + 16 -> 1 layer the gradient w.r.t. its up-sampled 16-channel source. */
size_t avsi_conv2d_thin_wgrad_workspace_bytes(int C0, int C1, int k, int Cout, int B, int H, int W);
int avsi_conv2d_thin_wgrad_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1,
                               int B, int H, int W, int k, const float* dy, int ldy, int Cout, float* dw, int ldw,
                               void* workspace, size_t workspace_bytes, void* stream);
int avsi_conv2d_thin_dx_coarse_f32(const float* dy, int ldy, const float* filter, int ldf, float* dsrc1_coarse,
                                   int ld1, int accumulate, int B, int H, int W, void* stream);
int avsi_im2col_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1,
                    int B, int H, int W, int k, float* col, int Kc, void* stream);
int avsi_col2im_f32(const float* dcol, int Kc, float* dsrc0, int C0, int ld0, float* dsrc1_coarse, int C1, int ld1,
                    int B, int H, int W, int k, int accumulate0, int accumulate1, void* stream);
size_t avsi_unet_workspace_bytes(int C);
int avsi_colstats_f32(const float* x, int64_t R, int C, int ld, float eps, float* mean, float* rstd,
                      void* workspace, size_t workspace_bytes, void* stream);
/* Batch norm + activation + 2 x 2 max pooling of an encoder layer in one pass (unet_layers.py:6-20 followed by the
 * pooling): pooled [B*H/2*W/2][ld] = max over the window of act(bn(x)); y, when given, also receives
 * the full-resolution activation (training keeps it for the backward pass).  ld % 4 == 0, 16-byte aligned. */
int avsi_bn_act_pool_f32(const float* x, int B, int H, int W, int C, int ld, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, int act, float* y, float* pooled, void* stream);
/* Convolution (operands as avsi_conv2d_f32) + the batch statistics of its output, tf.layers.batch_normalization(training=True)
 * (unet_layers.py:14,33): mean[Cout] and rstd[Cout] = 1 / sqrt(var + eps) come from partial sums the convolution's own
 * epilogue leaves per output tile (deterministic order) and one small finishing launch -- no pass over the output.  Takes the
 * 16-wide-MFMA route where avsi_conv2d_thin_mfma_supported says so, the implicit GEMM otherwise (same support rules).
 * src1_bn (NULL, or four device pointers mean, rstd, gamma, beta of C1 floats): src1_coarse is then the RAW convolution output
 * of the layer below and its batch norm + LeakyReLU(0.2) are applied while the operand is staged -- that layer's own
 * normalise / activate pass never runs (16-wide-MFMA route, k = 3, only: AVSI_ERR_UNSUPPORTED otherwise). */
size_t avsi_conv2d_bn_workspace_bytes(int B, int H, int W, int k, int C0, int C1, int Cout);
int avsi_conv2d_bn_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1,
                       const float* const* src1_bn, int B, int H, int W, int k, const float* filter, int ldf,
                       const float* bias, int Cout, float* out, int ldo, const float* zeros64, float eps, float* mean,
                       float* rstd, void* workspace, size_t workspace_bytes, void* stream);
/* Few-channel convolutions at high resolution on the 16-wide fp32 MFMA, input patch and filter in LDS (same operands as
 * avsi_conv2d_f32): the U-Net's decoder layer 16 + 32 -> 16 (3 x 3) and encoder layer 16 -> 32 (5 x 5); H % 4 == 0,
 * W % 32 == 0.  avsi_conv2d_thin_mfma_supported returns 1 for the shapes it takes. */
int avsi_conv2d_thin_mfma_supported(int k, int C0, int C1, int Cout, int H, int W);
/* avsi_conv2d_thin_mfma_f32 itself (no statistics, no deferred batch norm) also takes the input-gradient convolutions of those
 * two layers -- (k, C0, C1, Cout) = (5, 32, 0, 16) and (3, 16, 0, 48) with tap-flipped transposed filters: 1 for every shape it takes. */
int avsi_conv2d_thin_mfma_plain_supported(int k, int C0, int C1, int Cout, int H, int W);
int avsi_conv2d_thin_mfma_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H, int W,
                              int k, const float* filter, int ldf, const float* bias, int Cout, float* out, int ldo,
                              const float* zeros64, void* stream);
/* Filter gradient of the same few-channel layers (and of the 32 + 64 -> 32 decoder layer) on the 16-wide MFMA: the tile's
 * input patch and its dY go through LDS once, the filter gradient accumulates in registers over all the tiles of a workgroup,
 * one partial filter per workgroup in the workspace, summed in order (deterministic).  dw [k*k*(C0+C1)][Cout], ldw == Cout.
 * Same results as avsi_conv2d_wgrad_f32 up to summation order (reference: tf.gradients of unet_layers.py:11,31). */
int avsi_conv2d_thin_mfma_wgrad_supported(int k, int C0, int C1, int Cout, int H, int W);
size_t avsi_conv2d_thin_mfma_wgrad_workspace_bytes(int C0, int C1, int k, int Cout, int B, int H, int W);
int avsi_conv2d_thin_mfma_wgrad_f32(const float* src0, int C0, int ld0, const float* src1_coarse, int C1, int ld1, int B, int H,
                                    int W, int k, const float* dy, int ldy, int Cout, float* dw, int ldw, const float* zeros64,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* The first encoder layer at inference: 7 x 7 convolution of the one-channel input, bias, ReLU and 2 x 2 max pooling
 * fused (the full-resolution 16-channel activation is never written). */
int avsi_conv2d_thin_relu_pool_f32(const float* src0, int ld0, int B, int H, int W, int k, const float* filter, int ldf,
                                   const float* bias, int Cout, float* out, int ldo, void* stream);
int avsi_bn_act_f32(const float* x, int64_t R, int C, int ld, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, int act, float* y, void* stream);
int avsi_bn_act_bwd_f32(const float* x, const float* dy, int64_t R, int C, int ld, const float* mean,
                        const float* rstd, const float* gamma, const float* beta, int act, float* dx,
                        float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* Backward of an encoder layer whose activation act(bn(x)) was max-pooled 2 x 2 (avsi_bn_act_pool_f32; reference
 * unet_layers.py:6-20 + models.py:592-597 under tf.gradients): dpooled [B][H/2][W/2][ld] is the gradient of the POOLED output,
 * x [B][H][W][ld] the convolution output.  The window's activations are recomputed from x (same arithmetic as the forward
 * pass, first maximum wins ties), so neither the full-resolution activation nor its gradient exists in memory:
 *   dx = d loss / d x; dgamma, dbeta as avsi_bn_act_bwd_f32 (mean == null: no batch norm, dx = routed gradient * act');
 *   dbias (optional) = column sums of dx, the bias gradient of a layer without batch norm.
 * workspace: avsi_unet_workspace_bytes(C).  ld % 4 == 0, ld <= 256, 16-byte aligned operands. */
int avsi_bn_act_pool_bwd_f32(const float* x, const float* dpooled, int B, int H, int W, int C, int ld, const float* mean,
                             const float* rstd, const float* gamma, const float* beta, int act, float* dx, float* dgamma,
                             float* dbeta, float* dbias, void* workspace, size_t workspace_bytes, void* stream);
int avsi_maxpool2_f32(const float* x, float* y, int B, int H, int W, int C, int ld, void* stream);
int avsi_maxpool2_bwd_f32(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int ld, void* stream);

/* Stand-alone forms of the small operators of audio_processing.py (the inpainter itself uses the
 * fused front end).  avsi_spectrogram_f32: out[i] = |stft[i]|^power, log(. + eps) if do_log
 * (:45-56); avsi_logmel_f32: out[r][m] = log(sum_j spec[r][start[m]+j] w[m][j] + eps) (:59-72);
 * avsi_preemphasis_f32: y[b][t] = x[b][t] - alpha x[b][t-1] (:19-22); avsi_delta_f32: regression
 * deltas over [B][T][F] with window N, SYMMETRIC padding applied one frame at a time (:85-94). */
int avsi_spectrogram_f32(const float* stft, float* out, int64_t n, float power, int do_log, float eps, void* stream);
int avsi_logmel_f32(const float* spec, int64_t ld, int64_t rows, int num_mel, const int32_t* mel_start,
                    const int32_t* mel_len, const float* mel_w, int mel_w_stride, float* out, float eps, void* stream);
int avsi_preemphasis_f32(const float* x, float* y, int64_t B, int64_t N, int64_t ldx, float alpha, void* stream);
int avsi_delta_f32(const float* x, float* y, int64_t B, int T, int F, int N, void* stream);

/* Host helper (no GPU work): CRC-32C of a buffer, continuing from `seed` (0 to start).  The
 * TFRecord framing the reference's datasets use (tf.data.TFRecordDataset, dataset_reader.py:24)
 * stores masked CRC-32C values of the length and payload of every record. */
uint32_t avsi_crc32c(const void* data, size_t n, uint32_t seed);

/* Host helper (no GPU work): the WAV files of the inference driver, `wavfile.write(path, 16000,
 * enhanced[:seq_len * 192].astype(np.int16))` (inference.py:159-162), for `count` utterances: samples [count][stride]
 * float32 (host memory), num_samples[i] of row i go to paths[i] as 16-bit PCM mono with the 44-byte header
 * scipy.io.wavfile.write produces; float -> int16 as numpy.astype on x86 (truncation toward zero, low 16 bits).
 * make_dirs: create missing parent directories (os.makedirs(..., exist_ok=True)).  Thread-safe; the driver calls it
 * from a few host threads outside the interpreter lock.  AVSI_ERR_INVALID_ARG if a file cannot be written. */
int avsi_wav_write_batch_int16_host(const char* const* paths, const float* samples, int64_t stride,
                                    const int32_t* num_samples, int count, int sample_rate, int make_dirs);

/* Host helpers (no GPU work; HOST pointers; re-entrant): the native half of the TFRecord reader.  The
 * reference parses its samples inside TensorFlow's C++ input pipeline
 * (tf.parse_single_sequence_example over the 'fixed' schema, dataset_reader.py:62-79, with the
 * `embedding` context feature in dataset_reader_emb.py:63-81; writer tfrecord_utils.py:19-41).
 * avsi_sequence_example_shape_host: shape5 = {target_audio_wav samples, embedding size, mask frames,
 *   video_features frames, labels entries} of one serialized tf.train.SequenceExample.
 * avsi_sequence_example_decode_fixed_host: parses one record straight into (one row of) the caller's
 *   batch arrays: lengths2 = {sequence_length, labels_length}; wav_i32 [num_audio_samples] = the float
 *   samples truncated toward zero (tf.to_int32); embedding [embedding_size] (0 = not read); sample_path
 *   as a NUL-terminated string; labels [num_labels]; video [num_video_frames][video_feat_size];
 *   mask [num_frames][audio_feat_size].  AVSI_ERR_INVALID_ARG for a malformed record or a missing
 *   feature, AVSI_ERR_UNSUPPORTED when its sizes differ from the ones passed (ragged batch). */
int avsi_sequence_example_shape_host(const void* buf, size_t n, int64_t* shape5);
int avsi_sequence_example_decode_fixed_host(const void* buf, size_t n, int num_audio_samples,
                                            int audio_feat_size, int video_feat_size, int embedding_size,
                                            int num_frames, int num_video_frames, int num_labels,
                                            int32_t* lengths2, int32_t* wav_i32, float* embedding,
                                            char* sample_path, int sample_path_cap, float* labels,
                                            float* video, float* mask);

/* The same for a whole .tfrecord file that holds ONE record (the reference's datasets: one sample per file): open,
 * read, check the TFRecord framing (masked CRC-32C of length and payload; verify = 0 skips the checks), parse.  All of
 * it outside any interpreter lock, so readers may run files on parallel threads.  AVSI_ERR_INVALID_ARG: unreadable,
 * truncated, checksum mismatch or malformed record; AVSI_ERR_UNSUPPORTED: more than one record in the file, or sizes
 * that differ from the ones passed. */
int avsi_tfrecord_file_shape_host(const char* path, int verify, int64_t* shape5);
int avsi_tfrecord_file_decode_fixed_host(const char* path, int verify, int num_audio_samples, int audio_feat_size,
                                         int video_feat_size, int embedding_size, int num_frames,
                                         int num_video_frames, int num_labels, int32_t* lengths2, int32_t* wav_i32,
                                         float* embedding, char* sample_path, int sample_path_cap, float* labels,
                                         float* video, float* mask);
/* `count` such files in one call: row i of every output array ([count][...], contiguous) belongs to paths[i], codes[i] is
 * that file's status, the return value the first non-zero one.  The reader hands each of its threads one slice of a batch. */
int avsi_tfrecord_files_decode_fixed_host(const char* const* paths, int count, int verify, int num_audio_samples,
                                          int audio_feat_size, int video_feat_size, int embedding_size, int num_frames,
                                          int num_video_frames, int num_labels, int32_t* lengths2, int32_t* wav_i32,
                                          float* embedding, char* sample_paths, int sample_path_cap, float* labels,
                                          float* video, float* mask, int32_t* codes);

#ifdef __cplusplus
}
#endif
#endif /* AVSI_HIP_H */
