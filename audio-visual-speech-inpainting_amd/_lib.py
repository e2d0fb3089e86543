"""ctypes binding of libavsi_hip.so (the C ABI of include/avsi_hip.h).

Loading is lazy; ``lib()`` raises RuntimeError loudly when the shared library has not been
built (``make -C csrc`` or ``__graft_entry__.build()``) -- there is no fallback path.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libavsi_hip.so")
ABI_VERSION = 12

AVSI_OK = 0
AVSI_ERR_INVALID_ARG, AVSI_ERR_UNSUPPORTED, AVSI_ERR_LAUNCH, AVSI_ERR_WORKSPACE = -1, -2, -3, -4


class AvsiError(RuntimeError):
    pass


class FrontendArgs(Structure):
    """Mirror of ``avsi_frontend_args`` (include/avsi_hip.h)."""
    _fields_ = [
        ("wav", c_void_p), ("batch", c_int32), ("num_samples", c_int32), ("wav_stride", c_int64),
        ("frame_len", c_int32), ("hop", c_int32), ("nfft", c_int32), ("num_frames", c_int32),
        ("num_bins", c_int32), ("table", c_void_p),
        ("mean", c_void_p), ("stdev", c_void_p),
        ("mask", c_void_p), ("mask_stride_b", c_int64), ("mask_stride_t", c_int64),
        ("out_stft", c_void_p), ("stft_stride_b", c_int64), ("stft_stride_t", c_int64),
        ("out_spec", c_void_p), ("spec_stride_b", c_int64), ("spec_stride_t", c_int64),
        ("out_feat", c_void_p), ("feat_stride_b", c_int64), ("feat_stride_t", c_int64),
        ("feat_cols", c_int32),
        ("out_logmel", c_void_p), ("logmel_stride_b", c_int64), ("logmel_stride_t", c_int64),
        ("num_mel", c_int32), ("mel_start", c_void_p), ("mel_len", c_void_p), ("mel_w", c_void_p),
        ("mel_w_stride", c_int32),
        ("spec_power", c_float), ("log_spec", c_int32), ("eps", c_float),
    ]


class IstftArgs(Structure):
    """Mirror of ``avsi_istft_args`` (include/avsi_hip.h)."""
    _fields_ = [
        ("mode", c_int32),
        ("in0", c_void_p), ("in_stride_b", c_int64), ("in_stride_t", c_int64),
        ("in1", c_void_p), ("in1_stride_b", c_int64), ("in1_stride_t", c_int64),
        ("in2", c_void_p), ("in2_stride_b", c_int64), ("in2_stride_t", c_int64),
        ("mean", c_void_p), ("stdev", c_void_p),
        ("batch", c_int32), ("num_frames", c_int32), ("num_bins", c_int32), ("frame_len", c_int32),
        ("hop", c_int32), ("nfft", c_int32),
        ("table", c_void_p), ("out", c_void_p), ("out_stride_b", c_int64), ("num_samples", c_int64),
        ("wav", c_void_p), ("wav_stride_b", c_int64), ("wav_samples", c_int64),
    ]


class GemmEpilogue(Structure):
    """Mirror of ``avsi_gemm_epilogue`` (include/avsi_hip.h)."""
    _fields_ = [("bias", c_void_p), ("row_scale", c_void_p),
                ("row_map_bp", c_int32), ("row_map_t", c_int32), ("row_map_b", c_int32), ("k_zero", c_int32 * 4)]


# name -> (restype, argtypes); every symbol include/avsi_hip.h declares
PROTOTYPES = {
    "avsi_abi_version": (c_int, []),
    "avsi_status_string": (c_char_p, [c_int]),
    "avsi_blstm_net_supported": (c_int, [POINTER(c_int), c_int]),
    "avsi_blstm_rec_bwd_kernel_name": (c_char_p, [c_int]),
    "avsi_frontend_table_floats": (c_size_t, [c_int, c_int]),
    "avsi_frontend_init_tables": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "avsi_frontend_f32": (c_int, [POINTER(FrontendArgs), c_void_p]),
    "avsi_frontend_l1_loss_supported": (c_int, [POINTER(FrontendArgs)]),
    "avsi_frontend_l1_loss_f32": (c_int, [POINTER(FrontendArgs), c_void_p, c_int64, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_size_t,
                                          c_void_p]),
    "avsi_gemm_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_void_p, c_int64,
                              c_float, c_void_p, c_int64, POINTER(GemmEpilogue), c_void_p]),
    "avsi_pack_bf16x3_b_bytes": (c_size_t, [c_int, c_int]),
    "avsi_pack_bf16x3_b": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "avsi_gemm_bf16x3_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_void_p]),
    "avsi_dropout_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, ctypes.c_uint64, c_void_p]),
    "avsi_scale_elements_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "avsi_blstm_rec_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "avsi_blstm_rec_fwd_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "avsi_gemm_splitk_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "avsi_gemm_splitk_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_void_p, c_int64,
                                     c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "avsi_blstm_rec_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "avsi_relayout_rows_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                       c_int, c_void_p, c_int64, c_int64, c_void_p]),
    "avsi_colsum_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "avsi_colsum_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "avsi_adam_tf_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                 c_int64, c_float, c_float, c_void_p]),
    "avsi_step_guard_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "avsi_adam_tf_guarded_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                         c_int64, c_float, c_float, c_void_p, c_int, c_void_p]),
    "avsi_sgd_momentum_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_void_p,
                                      c_int, c_void_p]),
    "avsi_istft_table_floats": (c_size_t, [c_int, c_int, c_int]),
    "avsi_istft_init_tables": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "avsi_istft_f32": (c_int, [POINTER(IstftArgs), c_void_p]),
    "avsi_im2col_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                c_void_p]),
    "avsi_col2im_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_int, c_void_p]),
    "avsi_unet_workspace_bytes": (c_size_t, [c_int]),
    "avsi_colstats_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_size_t,
                                  c_void_p]),
    "avsi_bn_act_pool_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p, c_void_p]),
    "avsi_conv2d_bn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "avsi_conv2d_bn_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                   c_void_p, c_int, c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "avsi_conv2d_thin_mfma_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "avsi_conv2d_thin_mfma_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                          c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "avsi_conv2d_thin_relu_pool_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                               c_void_p, c_int, c_void_p]),
    "avsi_bn_act_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                c_void_p]),
    "avsi_bn_act_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "avsi_conv2d_thin_mfma_plain_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "avsi_conv2d_thin_mfma_wgrad_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "avsi_conv2d_thin_mfma_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "avsi_conv2d_thin_mfma_wgrad_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                                c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "avsi_bn_act_pool_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "avsi_maxpool2_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "avsi_maxpool2_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "avsi_spectrogram_f32": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_int, c_float, c_void_p]),
    "avsi_logmel_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                c_float, c_void_p]),
    "avsi_preemphasis_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p]),
    "avsi_delta_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "avsi_crc32c": (ctypes.c_uint32, [c_void_p, c_size_t, ctypes.c_uint32]),
    "avsi_l1_loss_workspace_bytes": (c_size_t, [c_int64]),
    "avsi_l1_loss_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p,
                                 c_size_t, c_void_p]),
    "avsi_blstm_rec_fwd_coop_workspace_bytes": (c_size_t, [c_int]),
    "avsi_blstm_rec_fwd_coop_exchange_bytes": (c_size_t, [c_int, c_int]),
    "avsi_blstm_rec_bwd_coop_exchange_bytes": (c_size_t, [c_int, c_int]),
    "avsi_blstm_rec_fwd_coop_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                            c_size_t, c_void_p]),
    "avsi_blstm_rec_fwd_coop_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                 c_void_p, c_size_t, c_void_p]),
    "avsi_blstm_rec_bwd_coop_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                            c_size_t, c_void_p]),
    "avsi_blstm_rec_fwd_cs_workspace_bytes": (c_size_t, [c_int]),
    "avsi_blstm_rec_fwd_cs_groups_per_launch": (c_int, [c_int, c_int, c_int]),
    "avsi_blstm_rec_fwd_cs_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                          c_size_t, c_void_p]),
    "avsi_blstm_rec_fwd_cs_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                               c_void_p, c_size_t, c_void_p]),
    "avsi_diag_cs_stamps": (c_int, [c_void_p]),
    "avsi_stream_delay_us": (c_int, [c_int, c_void_p]),
    "avsi_diag_occupy_cus": (c_int, [c_int, c_void_p, c_int, c_void_p]),
    "avsi_diag_copy_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "avsi_conv2d_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "avsi_conv2d_splitk_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "avsi_conv2d_splitk_suggest": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "avsi_conv2d_splitk_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                       c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "avsi_unet_tail_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "avsi_unet_tail_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                   c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_size_t, c_void_p]),
    "avsi_conv2d_thin_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                     c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "avsi_split_sumpool_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_void_p]),
    "avsi_conv2d_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "avsi_conv2d_wgrad_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                      c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "avsi_conv2d_thin_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "avsi_conv2d_thin_wgrad_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                           c_int, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "avsi_conv2d_thin_dx_coarse_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                               c_void_p]),
    "avsi_l1_loss_blend_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "avsi_ctc_loss_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "avsi_ctc_loss_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                  c_int, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "avsi_sequence_example_shape_host": (c_int, [c_void_p, c_size_t, c_void_p]),
    "avsi_sequence_example_decode_fixed_host": (c_int, [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                        c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                        c_void_p]),
    "avsi_tfrecord_file_shape_host": (c_int, [c_char_p, c_int, c_void_p]),
    "avsi_tfrecord_file_decode_fixed_host": (c_int, [c_char_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                     c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                     c_void_p]),
    "avsi_tfrecord_files_decode_fixed_host": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                      c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                      c_void_p, c_void_p]),
    "avsi_ctc_beam_search_host_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                              c_void_p, c_int, c_void_p, c_void_p]),
    "avsi_wav_write_batch_int16_host": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int]),
    "avsi_lws_num_frames": (c_int, [c_int, c_int, c_int]),
    "avsi_lws_table_floats": (c_size_t, [c_int, c_int, c_int]),
    "avsi_lws_init_tables": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "avsi_lws_stft_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "avsi_lws_stitch_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p]),
    "avsi_lws_run_workspace_bytes": (c_size_t, [c_int]),
    "avsi_lws_run_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_int,
                                 c_float, c_float, c_float, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "avsi_lws_run_skew_workspace_bytes": (c_size_t, [c_int, c_int]),
    "avsi_lws_skew_launch_shape": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "avsi_lws_run_skew_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_int,
                                      c_float, c_float, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "avsi_lws_run_duo_workspace_bytes": (c_size_t, [c_int, c_int]),
    "avsi_lws_run_duo_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_int,
                                      c_float, c_float, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "avsi_lws_istft_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "avsi_lws_istft_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int64, c_int, c_void_p,
                                   c_size_t, c_void_p]),
}

_lib = None


def lib():
    """Return the loaded CDLL; raise if it is not built or its ABI version mismatches."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise AvsiError(
                "libavsi_hip.so not found at %s: build it with `make -C %s` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
                % (LIB_PATH, os.path.dirname(LIB_PATH)))
        # PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  Its streams
        # and device pointers belong to THAT runtime instance, so it must be the one already
        # loaded when libavsi_hip.so's NEEDED entry is resolved: import torch first.
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(handle, name)      # AttributeError if a declared symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        got = handle.avsi_abi_version()
        if got != ABI_VERSION:
            raise AvsiError("libavsi_hip.so ABI version %d, host layer expects %d: rebuild" % (got, ABI_VERSION))
        _lib = handle
    return _lib


def check(status, what):
    if status != AVSI_OK:
        msg = lib().avsi_status_string(status)
        raise AvsiError("%s failed: %s (%d)" % (what, msg.decode() if msg else "?", status))


def require_cuda(*tensors):
    """Every compute entry point goes through this: device tensors only, no fallback."""
    import torch
    if not torch.cuda.is_available():
        raise AvsiError("no GPU visible: the avsi_amd hot path runs on MI355X only (no CPU fallback)")
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise AvsiError("expected a device tensor, got %s" % (t.device,))


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
