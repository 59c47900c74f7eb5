"""ctypes view of include/lstm_unet_hip.h: struct layouts and prototypes.  Pure declarations --
no policy, no fallbacks.  `bind(path)` loads ONE shared object and attaches the prototypes."""
import ctypes as C

ABI_VERSION = 12         # lu_abi_version() of include/lstm_unet_hip.h this binding was written for
c_f32p = C.c_void_p      # raw device pointers travel as integers
i32, i64, f32, f64 = C.c_int32, C.c_int64, C.c_float, C.c_double

LU_EPI_BIAS, LU_EPI_LSTM = 0, 1
LU_F32, LU_BF16 = 0, 1
# lu_conv_desc.flags / lu_wgrad_desc.flags (include/lstm_unet_hip.h)
LU_CONV_F_PATCH8, LU_CONV_F_PATCH16, LU_CONV_F_NO_HALO, LU_CONV_F_XCD_BY_N = 1, 2, 4, 8
LU_CONV_F_GENERAL = 64
LU_CONV_F_GATES_BF16, LU_CONV_F_SRC1_CENTER, LU_CONV_F_NO_BALANCE = 256, 512, 1024
LU_CONV_F_SLABS_ONLY = 2048
LU_CONV_F_NO_NARROW = 4096
LU_CONV_F_HALF_BLOCK = 8192
LU_CONV_F_H16_SPLIT = 65536
LU_WGRAD_F_NO_ROW, LU_WGRAD_F_NO_SMALL3, LU_WGRAD_F_CT64, LU_WGRAD_F_CT128, LU_WGRAD_F_SMALL_TILE, LU_WGRAD_F_PRB32 = 1, 2, 4, 8, 16, 32
LU_WGRAD_F_NO_RAGGED = 64
LU_WGRAD_F_NO_NARROW_BF16 = 128
LU_WGRAD_F_NO_TAPS9 = 2048
LU_WGRAD_F_NO_DMA = 8192
LU_WGRAD_F_PIECES3 = 131072


class ConvSrc(C.Structure):
    _fields_ = [('x', c_f32p), ('w', c_f32p), ('frame_stride', i64), ('w_tap_stride', i64),
                ('pix_stride', i32), ('C', i32), ('w_row_stride', i32), ('dtype', i32)]


class ConvDesc(C.Structure):
    _fields_ = [('src', ConvSrc * 2), ('n_src', i32), ('frames', i32), ('Hin', i32), ('Win', i32),
                ('Hout', i32), ('Wout', i32), ('k', i32), ('stride', i32), ('dil', i32),
                ('pad_t', i32), ('pad_l', i32), ('N', i32), ('out_pix_stride', i32), ('epilogue', i32),
                ('bias', c_f32p), ('out', c_f32p), ('out_frame_stride', i64),
                ('c_prev', c_f32p), ('c_out', c_f32p), ('h_out', c_f32p), ('gates_out', c_f32p),
                ('c_prev_frame_stride', i64), ('c_out_frame_stride', i64), ('h_frame_stride', i64),
                ('gates_frame_stride', i64), ('splits', i32), ('precision', i32), ('workspace', C.c_void_p),
                ('out_row_stride', i64), ('k_h', i32), ('flags', i32), ('h16_out', C.c_void_p), ('h16_frame_stride', i64),
                ('post_scale', c_f32p), ('post_shift', c_f32p), ('post_alpha', f32)]


class WgradDesc(C.Structure):
    _fields_ = [('x', c_f32p), ('x_frame_stride', i64), ('x_pix_stride', i32), ('C', i32),
                ('dy', c_f32p), ('dy_frame_stride', i64), ('dy_pix_stride', i32), ('N', i32),
                ('frames', i32), ('Hin', i32), ('Win', i32), ('Hout', i32), ('Wout', i32),
                ('k', i32), ('stride', i32), ('pad_t', i32), ('pad_l', i32),
                ('dw', c_f32p), ('dw_tap_stride', i64), ('dw_row_stride', i32), ('splits', i32),
                ('beta', f32), ('precision', i32), ('workspace', C.c_void_p),
                ('dbias', c_f32p), ('dbias_beta', f32), ('phase', i32), ('x_dtype', i32), ('dy_dtype', i32), ('flags', i32),
                ('terms', i32), ('x_term_stride', i64), ('dy_term_stride', i64)]


class PrepOp(C.Structure):      # lu_prep_op: one flip (kind 0) / bf16 pack (kind 1) of lu_weight_prep_batch's device table
    _fields_ = [('kind', i32), ('blk0', i32), ('nblk', i32), ('k', i32), ('src', c_f32p), ('dst', C.c_void_p),
                ('tap_stride', i64), ('row_stride', i32), ('kk', i32), ('C', i32), ('N', i32), ('C_tot', i32), ('c_off', i32)]


P = C.c_void_p
S = C.c_void_p  # stream
PROTOTYPES = {
    'lu_last_error': (C.c_char_p, []),
    'lu_abi_version': (C.c_int, []),
    'lu_conv2d_fwd': (C.c_int, [C.POINTER(ConvDesc), S]),
    'lu_pack_weights_taps_bf16': (C.c_int, [P, i64, C.c_int, C.c_int, C.c_int, C.c_int, P, S]),
    'lu_pack_weights_bf16_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'lu_pack_weights_bf16': (C.c_int, [P, i64, C.c_int, C.c_int, C.c_int, C.c_int, P, S]),
    'lu_pack_weights_split6_bf16': (C.c_int, [P, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, S]),
    'lu_conv2d_workspace_bytes': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'lu_stride2_dgrad_weights': (C.c_int, [P, P] + [C.c_int] * 10 + [S]),
    'lu_weight_flip_transpose': (C.c_int, [P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, S]),
    'lu_weight_prep_batch': (C.c_int, [P, C.c_int, C.c_int, S]),
    'lu_conv2d_wgrad_workspace_bytes': (C.c_size_t, [C.POINTER(WgradDesc)]),
    'lu_conv2d_wgrad': (C.c_int, [C.POINTER(WgradDesc), S]),
    'lu_lstm_gates_fwd': (C.c_int, [P, P, P, P, P, i32, i64, i32, i64, S]),
    'lu_lstm_gates_fwd_slabs': (C.c_int, [P, i32, P, P, P, P, P, i32, i64, i32, i64, S]),
    'lu_lstm_gates_bwd': (C.c_int, [P, P, P, P, i64, P, P, P, P, i32, i64, i32, S]),
    'lu_lstm_gates_bwd_bf16': (C.c_int, [P, P, P, P, i64, P, P, P, i32, i64, i32, S]),
    'lu_lstm_gates_bwd_split': (C.c_int, [P, P, P, P, i64, P, P, P, P, i32, i64, i32, S]),
    'lu_convert_f32_bf16': (C.c_int, [P, P, i64, S]),
    'lu_split6': (C.c_int, [P, i64, i32, i64, P, i64, i32, i32, i32, S]),
    'lu_convert_bf16_f32': (C.c_int, [P, P, i64, S]),
    'lu_im2col_bf16': (C.c_int, [P, P, i32, i32, i32, i32, i32, S]),
    'lu_colreduce_workspace_bytes': (C.c_size_t, [i64, i32]),
    'lu_colsum': (C.c_int, [P, i64, i32, i32, P, f32, P, S]),
    'lu_bn_stats': (C.c_int, [P, i64, i32, P, P, S]),
    'lu_bn_finalize_train': (C.c_int, [P, f64, P, P, f32, f32, P, P, P, P, P, P, i32, S]),
    'lu_bn_finalize_infer': (C.c_int, [P, P, P, P, f32, P, P, i32, S]),
    'lu_bn_lrelu_apply': (C.c_int, [P, P, P, P, f32, i64, i32, S]),
    'lu_bn_lrelu_bwd_reduce': (C.c_int, [P, P, P, P, P, P, f32, i64, i32, P, P, S]),
    'lu_bn_lrelu_bwd_apply': (C.c_int, [P, P, P, P, P, P, f32, P, f64, P, P, P, i64, i32, S]),
    'lu_bn_lrelu_bwd_apply_bf16': (C.c_int, [P, P, P, P, P, P, f32, P, f64, P, P, P, i64, i32, S]),
    'lu_upsample2x_fwd': (C.c_int, [P, P, i32, i32, i32, i32, i32, S]),
    'lu_upsample2x_bwd': (C.c_int, [P, i32, P, i32, i32, i32, i32, i32, S]),
    'lu_window_copy': (C.c_int, [P, i32, P, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, S]),
    'lu_wce_workspace_bytes': (C.c_size_t, [i64]),
    'lu_softmax_wce_fwd': (C.c_int, [P, P, P, P, P, i64, P, S]),
    'lu_softmax_wce_bwd': (C.c_int, [P, P, P, P, f32, P, i64, S]),
    'lu_wce_finalize': (C.c_int, [P, P, S]),
    'lu_softmax3': (C.c_int, [P, P, i64, S]),
    'lu_softmax_rows': (C.c_int, [P, P, i64, C.c_int32, S]),
    'lu_adam_step': (C.c_int, [P, P, P, P, i64, f32, f32, f32, f32, f32, S]),
    'lu_scale_frames': (C.c_int, [P, P, i32, i64, S]),
    'lu_state_begin': (C.c_int, [P, P, P, P, i32, i64, S]),
    'lu_transpose_inner': (C.c_int, [P, P, i64, i32, i32, S]),
    'lu_add_inplace': (C.c_int, [P, P, i64, S]),
    'lu_crc32c': (C.c_uint32, [P, C.c_size_t, C.c_uint32]),
    'lu_conv2d_s2_fwd_bf16': (C.c_int, [P, i64, i32, P, P, i32, i32, i32, i32, i32, P, S]),
    'lu_conv2d_s2_dgrad_bf16': (C.c_int, [P, i64, i32, P, i32, i32, i32, i32, i32, P, S]),
    'lu_upsample2x_fwd_bf16': (C.c_int, [P, P, i32, i32, i32, i32, i32, S]),
    'lu_bn_lrelu_apply_bf16': (C.c_int, [P, P, P, P, f32, i64, i32, S]),
    'lu_post_workspace_bytes': (C.c_size_t, [i32, i32]),
    'lu_post_max_labels': (i32, [i32, i32]),
    'lu_post_label': (C.c_int, [P, i32, i32, f32, f64, P, P, P, P, S]),
    'lu_post_label_stats': (C.c_int, [P, i32, i32, i32, P, P, P, P, S]),
    'lu_post_fill_object': (C.c_int, [P, i32, i32, i32, i32, i32, i32, i32, P, P, S]),
    'lu_post_fill_all': (C.c_int, [P, i32, i32, P, P, P, P, P, S]),
    'lu_post_newid': (C.c_int, [P, P, P, i32, i32, i32, P, P, P, S]),
    'lu_post_frame': (C.c_int, [P, i32, i32, f32, f64, i32, i32, i32, i32, P, P, P, P, P, P, P, S]),
    'lu_post_frame_tail': (C.c_int, [i32, i32, i32, i32, i32, i32, P, P, P, P, P, S]),
    'lu_post_bbox_of_label': (C.c_int, [P, i32, i32, i32, P, S]),
    'lu_post_present': (C.c_int, [P, i32, i32, i32, i32, i32, P, S]),
    'lu_post_relabel': (C.c_int, [P, i32, i32, P, i32, P, S]),
}


def bind(path):
    """dlopen `path` and attach every prototype; raises if a declared symbol is missing."""
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    return lib
