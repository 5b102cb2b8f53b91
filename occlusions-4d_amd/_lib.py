"""ctypes binding of libocc4d.so (include/occ4d.h).  Fails loudly when the library is
missing or stale: the product has no CPU / PyTorch fallback path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libocc4d.so')
ABI_VERSION = 4

OK, EINVAL, ELAUNCH = 0, -1, -2

_f = C.c_void_p      # device float*
_i = C.c_void_p      # device int32*/int64*
_s = C.c_void_p      # hipStream_t


class LinearArgs(C.Structure):
    """occ4d_linear_args (include/occ4d.h)."""
    _fields_ = [
        ('x', C.c_void_p), ('ldx', C.c_int64),
        ('w', C.c_void_p), ('ldw', C.c_int64),
        ('bias', C.c_void_p),
        ('residual', C.c_void_p), ('ldr', C.c_int64),
        ('y', C.c_void_p), ('ldy', C.c_int64),
        ('M', C.c_int32), ('K', C.c_int32), ('N', C.c_int32),
        ('relu_in', C.c_int32), ('relu_out', C.c_int32),
        ('add_rows', C.c_void_p), ('ld_add', C.c_int64), ('add_div', C.c_int32),
        ('sub_rows', C.c_void_p), ('ld_sub', C.c_int64), ('sub_idx', C.c_void_p),
    ]


# path-level entry points (include/occ4d.h, last section)
PATH_DEFAULT, PATH_UNFUSED, PATH_FIRST_GEN, PATH_GENERIC_LINEAR, PATH_TRUNK4, PATH_FUSED_INTERP, PATH_BF16X6, PATH_BF16X6_TRUNK = 0, 1, 2, 8, 16, 32, 64, 128
PATH_SPLIT_F16 = 256
PROFILE_CROSS_ATTN, PROFILE_RESBLOCK, PROFILE_ROWLIN = 1, 2, 3
MAX_BLOCKS, MAX_CROSS = 16, 4
PROFILE_KINDS = {'cross_attn': PROFILE_CROSS_ATTN, 'resblock': PROFILE_RESBLOCK, 'rowlin': PROFILE_ROWLIN}


class PtLayerWeights(C.Structure):
    """occ4d_pt_layer_weights (include/occ4d.h)."""
    _fields_ = [(n, C.c_int32) for n in ('dim', 'dim2', 'pos_hidden', 'cross', 'd_in', 'd_out', 'reserved0', 'reserved1')] + \
               [(n, C.c_void_p) for n in ('to_q', 'to_k', 'to_v', 'pos0_w', 'pos0_b', 'pos2_w', 'pos2_b', 'attn0_w',
                                          'attn0_b', 'attn2_w', 'attn2_b', 'pre_w', 'pre_b', 'post_w', 'post_b')]


class LaunchEvents(C.Structure):
    """occ4d_launch_events (include/occ4d.h)."""
    _fields_ = [('events', C.POINTER(C.c_void_p)), ('capacity', C.c_int32), ('used', C.c_int32), ('kernel', C.c_int32),
                ('reserved', C.c_int32)]


class DecoderWeights(C.Structure):
    """occ4d_decoder_weights (include/occ4d.h)."""
    _fields_ = [(n, C.c_int32) for n in ('d_in', 'n_freq', 'd_hidden', 'd_out', 'd_latent', 'd_latent_local', 'n_blocks',
                                         'n_cross', 'k_local', 'k_cross', 'activation', 'lin_in_ld')] + \
               [('base_frequency', C.c_float), ('reserved', C.c_float)] + \
               [(n, C.c_void_p) for n in ('lin_in_w', 'lin_in_b', 'lin_out_w', 'lin_out_b')] + \
               [(n, C.c_void_p * MAX_BLOCKS) for n in ('lin_z_w', 'lin_z_b', 'fc0_w', 'fc0_b', 'fc1_w', 'fc1_b')] + \
               [('cross_after', C.c_int32 * MAX_CROSS), ('cross', PtLayerWeights * MAX_CROSS)]


_LW, _DW, _EV = C.POINTER(PtLayerWeights), C.POINTER(DecoderWeights), C.POINTER(LaunchEvents)

# name -> (restype, argtypes): every symbol include/occ4d.h declares
SIGNATURES = {
    'occ4d_abi_version': (C.c_int, []),
    'occ4d_is_cpu_twin': (C.c_int, []),
    'occ4d_last_error': (C.c_char_p, []),
    'occ4d_knn_f32': (C.c_int, [_f, C.c_int64, C.c_int, _f, C.c_int64, C.c_int, C.c_int, C.c_int, _i, C.c_int,
                                _f, _s]),
    'occ4d_fps_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, _i, _i, _s]),
    'occ4d_fps_start_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, C.c_int, _i, _i, _s]),
    'occ4d_fps_coop_workspace_bytes': (C.c_int64, []),
    'occ4d_fps_coop_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _i, _i, _s, _s]),
    'occ4d_copy_rows_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, C.c_int, C.c_int, _s]),
    'occ4d_fill_rows_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, C.c_float, _s]),
    'occ4d_nested_fps_level_i32': (C.c_int, [_i, _i, C.c_int, C.c_int, _i, _i, _s]),
    'occ4d_fps_repair_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, C.c_int, _i, _i, _s, _s]),
    'occ4d_fps_coop_debug': (C.c_int, [C.c_uint, C.c_int]),
    'occ4d_linear_f32': (C.c_int, [C.POINTER(LinearArgs), _s]),
    'occ4d_pt_pos_hidden_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _i, C.c_int, C.c_int, _f, _f, C.c_int,
                                          _f, _s]),
    'occ4d_pt_attn_in_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _i, C.c_int, C.c_int, C.c_int, _f, _s]),
    'occ4d_pt_softmax_agg_f32': (C.c_int, [_f, _f, C.c_int64, _f, _i, C.c_int, C.c_int, C.c_int, C.c_float, _f,
                                           C.c_int64, _s]),
    'occ4d_pt_cross_attn_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, C.c_int64, _i, _f, C.c_int64, _f,
                                          C.c_int64, _f, _f, _f, _f, _f, _f, _f, _f, C.c_int64, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_float, _s]),
    'occ4d_pt_self_attn16_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, C.c_int64, _i, _f, C.c_int64, _f,
                                           C.c_int64, _f, _f, _f, _f, _f, _f, _f, C.c_int64, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_float, _s]),
    'occ4d_matmul_f64': (C.c_int, [_f, C.c_int64, C.c_int64, _f, C.c_int64, C.c_int64, _f, C.c_int, C.c_int, C.c_int, _s]),
    'occ4d_pt_cross_attn16p_stream_floats': (C.c_int64, []),
    'occ4d_pt_cross_attn16p_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, C.c_int64, _i, _f, C.c_int64, _f,
                                             C.c_int64, _f, _f, _f, _f, C.c_int64, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_float, C.c_int, _s]),
    'occ4d_pt_cross_attn16p_logits_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, C.c_int64, _i, _f, C.c_int64, _f,
                                                    C.c_int64, _f, _f, _f, _f, C.c_int64, _f, _f, _f, _f, C.c_int, C.c_int,
                                                    C.c_int, C.c_int, C.c_float, C.c_int, _s]),
    'occ4d_pt_cross_attn_bf16x6_logits_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, C.c_int64, _i, _f, C.c_int64, _f,
                                                        C.c_int64, _f, _f, _f, _f, C.c_int64, _f, _f, _f, _f, C.c_int, C.c_int,
                                                        C.c_int, C.c_int, C.c_float, _s]),
    'occ4d_implicit_loss_workspace_floats': (C.c_int64, [C.c_int]),
    'occ4d_implicit_loss_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                          C.c_float, _f, _f, _f, C.c_int64, _s]),
    'occ4d_resblock_f16x3_packed_floats': (C.c_int64, []),
    'occ4d_pack_resblock_f16x3_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _s]),
    'occ4d_resblock_f16x3_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, _f, C.c_int, _s]),
    'occ4d_pt_pair_mlp_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _i, _f, _f, _f, _f, _f, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, _s]),
    'occ4d_layernorm_f32': (C.c_int, [_f, C.c_int64, _f, _f, C.c_float, C.c_int, _f, C.c_int64, C.c_int, C.c_int,
                                      _s]),
    'occ4d_maxpool_gather_f32': (C.c_int, [_f, C.c_int64, _i, C.c_int, C.c_int, C.c_int, _f, C.c_int64, _s]),
    'occ4d_gather_rows_f32': (C.c_int, [_f, C.c_int64, _i, C.c_int, C.c_int, _f, C.c_int64, _s]),
    'occ4d_mean_rows_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, _f, _s]),
    'occ4d_posenc_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, _f, C.c_int64, _s]),
    'occ4d_interp_weights_f32': (C.c_int, [_f, C.c_int, C.c_int, _f, _s]),
    'occ4d_interp_add_f32': (C.c_int, [_f, C.c_int64, _f, _f, C.c_int64, _i, _f, C.c_int, C.c_int, C.c_int, _s]),
    'occ4d_trunk_width': (C.c_int, []),
    'occ4d_trunk_packed_floats': (C.c_int64, [C.c_int]),
    'occ4d_resblock_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, _f, _f, _f, _f, C.c_int64, _i, _f, C.c_int,
                                     C.c_int, _s]),
    'occ4d_rowlin_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, C.c_int, C.c_int, _f, C.c_int64, _f, _f,
                                   C.c_int64, _i, _f, C.c_int, C.c_int, _s]),
    'occ4d_rowlin_masked_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, C.c_int, C.c_int, _f, C.c_int64, _f,
                                          C.c_int64, C.c_int, _s]),
    'occ4d_rowlin4_masked_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, C.c_int, C.c_int, _f, C.c_int64, _f,
                                          C.c_int64, C.c_int, _s]),
    'occ4d_rowlin4_masked_skip_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, C.c_int, C.c_int, _f, C.c_int64, _f,
                                               C.c_int64, C.c_int, _s]),
    'occ4d_trunk4_packed_floats': (C.c_int64, [C.c_int]),
    'occ4d_resblock4_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, _f, _f, _f, _f, C.c_int64, _i, _f, C.c_int,
                                      C.c_int, _s]),
    'occ4d_rowlin4_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, C.c_int, C.c_int, _f, C.c_int64, _f, _f,
                                    C.c_int64, _i, _f, C.c_int, C.c_int, _s]),
    'occ4d_squash_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int32), _s]),
    'occ4d_grid_points_f32': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_float, _f, _s]),
    'occ4d_split_count_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_float, _i, _i, _s]),
    'occ4d_radius_grid_workspace_bytes': (C.c_int64, [C.c_int]),
    'occ4d_radius_grid_build_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_float, _f, _s]),
    'occ4d_radius_far_f32': (C.c_int, [_f, C.c_int64, C.c_int, _f, C.c_float, _f, _s]),
    'occ4d_knn_grid_f32': (C.c_int, [_f, C.c_int64, C.c_int, _f, C.c_int64, C.c_int, C.c_int, C.c_int, _i, _f, _f, _s]),
    'occ4d_compact_count_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_float, C.c_int, _i, _i, _s]),
    'occ4d_compact_rows_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, _f, C.c_int64, C.c_float, C.c_int, _i, _f, _f,
                                         _s]),
    'occ4d_split_write_f32': (C.c_int, [_f, _f, C.c_int64, C.c_int, C.c_int, C.c_float, _i, C.c_int, C.c_int, _f, _f,
                                        _s]),
    # backward pass
    'occ4d_linear_wgrad_workspace': (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    'occ4d_linear_wgrad_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, C.c_int, C.c_int, C.c_int, _f, C.c_int, _f,
                                         C.c_int, _s]),
    'occ4d_linear_wgrad_bias_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _f, _f,
                                              C.c_int, _f, C.c_int, _s]),
    'occ4d_colsum_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, _f, C.c_int, _f, C.c_int, _s]),
    'occ4d_bn_workspace_doubles': (C.c_int64, [C.c_int, C.c_int]),
    'occ4d_bn_train_fwd_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, _f, _f, C.c_float, _f, _f, _f, C.c_int64, _f, _s]),
    'occ4d_bn_train_bwd_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, C.c_int64, C.c_int, C.c_int, _f, _f, _f, C.c_float, _f,
                                         C.c_int64, _f, _f, _f, _s]),
    'occ4d_swish_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, _f, C.c_int64, _s]),
    'occ4d_swish_bwd_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, C.c_int, C.c_int, _f, C.c_int64, _s]),
    'occ4d_relu_mask_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, C.c_int, C.c_int, _f, C.c_int64, _s]),
    'occ4d_scatter_add_rows_f32': (C.c_int, [_f, C.c_int64, _i, C.c_int, C.c_int, C.c_float, _f, C.c_int64, _s]),
    'occ4d_segment_sum_f32': (C.c_int, [_f, C.c_int, C.c_int, C.c_int, _f, C.c_int64, _s]),
    'occ4d_maxpool_gather_bwd_f32': (C.c_int, [_f, C.c_int64, _i, C.c_int, C.c_int, C.c_int, _f, C.c_int64, _f,
                                               C.c_int64, _s]),
    'occ4d_layernorm_bwd_f32': (C.c_int, [_f, C.c_int64, _f, _f, C.c_int64, C.c_float, C.c_int, C.c_int, _f,
                                          C.c_int64, _f, _f, _s]),
    'occ4d_pt_softmax_agg_bwd_f32': (C.c_int, [_f, _f, C.c_int64, _f, _i, C.c_int, C.c_int, C.c_int, C.c_float, _f,
                                               C.c_int64, _f, _f, _f, C.c_int64, _s]),
    'occ4d_pt_pos_hidden_bwd_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _i, C.c_int, C.c_int, C.c_int, _f, _f,
                                              _f, _f, _s]),
    'occ4d_interp_bwd_f32': (C.c_int, [_f, C.c_int64, _i, _f, C.c_int, C.c_int, C.c_int, _f, C.c_int64, _s]),
    'occ4d_segment_gather_sum_f32': (C.c_int, [_f, C.c_int64, _i, _i, _f, C.c_int, C.c_int, C.c_int, C.c_float, _f,
                                               C.c_int64, _s]),
    'occ4d_segments_workspace_ints': (C.c_int64, [C.c_int]),
    'occ4d_segments_build_i32': (C.c_int, [_i, C.c_int64, C.c_int, _i, _i, _i, _s]),
    'occ4d_segment_sum_sorted_f32': (C.c_int, [_f, C.c_int64, _i, _i, C.c_int, C.c_int, C.c_int, C.c_float, _f, C.c_int64,
                                               _s]),
    'occ4d_pt_pos_hidden_bwd_det_workspace': (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    'occ4d_pt_pos_hidden_bwd_det_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _i, C.c_int, C.c_int, C.c_int, _f, _f,
                                                  _f, _f, _f, _s]),
    'occ4d_axpby_f32': (C.c_int, [_f, C.c_int64, C.c_float, _f, C.c_int64, C.c_float, C.c_int, C.c_int, _f,
                                  C.c_int64, _s]),
    'occ4d_adamw_chunk': (C.c_int, []),
    'occ4d_adamw_clip_f32': (C.c_int, [_f, _f, _f, _i, _i, _i, C.c_int, _i, _i, C.c_int] + [C.c_float] * 6 + [_f, _s]),
    'occ4d_broadcast_rows_f32': (C.c_int, [_f, C.c_float, C.c_int, C.c_int, _f, C.c_int64, _s]),
    # packers + path-level entry points
    'occ4d_pack_trunk_rows_f32': (C.c_int, [_f, C.c_int64, C.c_int, _f, _s]),
    'occ4d_pack_trunk_cols_f32': (C.c_int, [_f, C.c_int64, _f, _s]),
    'occ4d_pack_trunk4_rows_f32': (C.c_int, [_f, C.c_int64, C.c_int, _f, _s]),
    'occ4d_pack_trunk4_cols_f32': (C.c_int, [_f, C.c_int64, _f, _s]),
    'occ4d_pack_attn16p_stream_f32': (C.c_int, [_f, _f, _f, _f, _s]),
    'occ4d_pt_cross_attn_bf16x6_stream_floats': (C.c_int64, []),
    'occ4d_debug_x6_stamps': (C.c_int, [C.c_void_p, C.c_int]),
    'occ4d_rowlin_bf16x6_packed_floats': (C.c_int64, [C.c_int]),
    'occ4d_pack_rowlin_bf16x6_f32': (C.c_int, [_f, C.c_int64, C.c_int, _f, _s]),
    'occ4d_rowlin_bf16x6_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, C.c_int, C.c_int, _f, C.c_int64, C.c_int, _s]),
    'occ4d_pt_pair_mlp_bf16x6_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _i, _f, _f, _f, _f, _f, C.c_int, C.c_int, C.c_int,
                                              C.c_int, _s]),
    'occ4d_rowlin_bf16x6_masked_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, C.c_int, C.c_int, _f, C.c_int64, C.c_int,
                                                _f, C.c_int64, C.c_int, _s]),
    'occ4d_pack_attn_bf16x6_stream_f32': (C.c_int, [_f, _f, _f, _f, _s]),
    'occ4d_pt_cross_attn_bf16x6_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, C.c_int64, _i, _f, C.c_int64, _f, C.c_int64,
                                                 _f, _f, _f, _f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _s]),
    'occ4d_pt_cross_attn_f16x3_stream_floats': (C.c_int64, []),
    'occ4d_pack_attn_f16x3_stream_f32': (C.c_int, [_f, _f, _f, _f, _s]),
    'occ4d_pt_cross_attn_f16x3_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, C.c_int64, _i, _f, C.c_int64, _f, C.c_int64,
                                                _f, _f, _f, _f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _s]),
    'occ4d_pt_cross_attn_f16x3_hidden_scale': (C.c_float, []),
    'occ4d_pt_cross_attn_f16x3_prescaled_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, C.c_int64, _i, _f, C.c_int64, _f,
                                                          C.c_int64, _f, _f, _f, _f, C.c_int64, C.c_int, C.c_int, C.c_int,
                                                          C.c_int, C.c_float, _s]),
    'occ4d_pt_cross_attn_f16w_stream_floats': (C.c_int64, []),
    'occ4d_pack_attn_f16w_stream_f32': (C.c_int, [_f, _f, _f, _f, _s]),
    'occ4d_pt_cross_attn_f16w_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, C.c_int64, _i, _f, C.c_int64, _f, C.c_int64,
                                               _f, _f, _f, _f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _s]),
    'occ4d_rowlin_f16x3_packed_floats': (C.c_int64, [C.c_int]),
    'occ4d_pack_rowlin_f16x3_f32': (C.c_int, [_f, C.c_int64, C.c_int, _f, _s]),
    'occ4d_rowlin_f16x3_f32': (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, C.c_int, C.c_int, _f, C.c_int64, C.c_int, _s]),
    'occ4d_pt_layer_prepared_floats': (C.c_int64, [_LW, C.c_int]),
    'occ4d_pt_layer_prepare_f32': (C.c_int, [_LW, _f, C.c_int, _s]),
    'occ4d_pt_layer_scene_floats': (C.c_int64, [_LW, C.c_int]),
    'occ4d_pt_layer_scene_f32': (C.c_int, [_LW, _f, _f, C.c_int64, C.c_int, _f, C.c_int, _s]),
    'occ4d_pt_layer_workspace_floats': (C.c_int64, [_LW, C.c_int, C.c_int, C.c_int, C.c_int]),
    'occ4d_pt_layer_fwd_f32': (C.c_int, [_LW, _f, _f, C.c_int64, _f, C.c_int64, C.c_int, _f, C.c_int64, _f, C.c_int64,
                                         C.c_int, C.c_int, _i, _f, _f, C.c_int64, _f, C.c_int, _EV, _s]),
    'occ4d_pt_layer_fwd_logits_f32': (C.c_int, [_LW, _f, _f, C.c_int64, _f, C.c_int64, C.c_int, _f, C.c_int64, _f, C.c_int64,
                                                C.c_int, C.c_int, _i, _f, _f, C.c_int64, _f, _f, _f, _f, C.c_int, _EV, _s]),
    'occ4d_down_pool_fwd_f32': (C.c_int, [_f, C.c_int64, C.c_int, C.c_int, _f, _f, C.c_int, C.c_int, _f, _f, _f, _f,
                                          C.c_float, _i, C.c_int, C.c_int, _f, C.c_int64, _f, _s]),
    'occ4d_decoder_prepared_floats': (C.c_int64, [_DW, C.c_int]),
    'occ4d_decoder_prepare_f32': (C.c_int, [_DW, _f, C.c_int, _s]),
    'occ4d_decoder_scene_floats': (C.c_int64, [_DW, C.c_int]),
    'occ4d_decoder_prepare_scene_f32': (C.c_int, [_DW, _f, _f, C.c_int64, _f, C.c_int64, _f, C.c_int, _f, C.c_int, _s]),
    'occ4d_decoder_query_workspace_floats': (C.c_int64, [_DW, C.c_int, C.c_int, C.c_int]),
    'occ4d_decoder_query_fwd_f32': (C.c_int, [_DW, _f, _f, C.c_int, _f, C.c_int64, C.c_int, _i, _i, _f, C.c_int64, _f,
                                              C.c_int64, _f, C.c_int, _EV, _s]),
    'occ4d_knn_dists_f32': (C.c_int, [_f, C.c_int64, C.c_int, _f, C.c_int64, C.c_int, _i, C.c_int, C.c_int, _f, _s]),
}

_lib = None
_twin = False            # True only after an explicit load_cpu_twin(): host pointers, no streams (cpu_twin.py)


class NativeLibraryError(RuntimeError):
    pass


def is_twin():
    return _twin


def _missing(name):
    def stub(*_a, **_k):
        raise NotImplementedError('%s is not part of the CPU twin (libocc4d_cpu.so holds the inference path only)' % name)
    return stub


class _TwinHandle:
    """The twin's exports with the argument types of SIGNATURES; an entry point the twin does not have raises
    NotImplementedError when CALLED (the HIP library, in contrast, must export every symbol of the header)."""

    def __init__(self, handle):
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                setattr(self, name, _missing(name))
                continue
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)


def load_cpu_twin(path):
    """Replaces the process's library handle by the g++ twin at `path`.  Only cpu_twin.enable() calls this; nothing in
    the package does so on its own (no fallback: without this call a missing libocc4d.so raises NativeLibraryError)."""
    global _lib, _twin
    handle = C.CDLL(path)
    twin = _TwinHandle(handle)
    if twin.occ4d_abi_version() != ABI_VERSION or twin.occ4d_is_cpu_twin() != 1:
        raise NativeLibraryError('%s is not the CPU twin of ABI version %d' % (path, ABI_VERSION))
    _lib, _twin = twin, True
    return twin


def unload_cpu_twin():
    global _lib, _twin
    _lib, _twin = None, False


def lib():
    """The loaded library (cached).  Raises NativeLibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            'libocc4d.so not found at %s -- build it with `python occlusions-4d_amd/build.py` '
            '(or __graft_entry__.build()); this package has no CPU/PyTorch fallback.' % LIB_PATH)
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise NativeLibraryError('libocc4d.so is stale: symbol %s missing; rebuild it' % name)
        fn.restype = res
        fn.argtypes = args
    if handle.occ4d_abi_version() != ABI_VERSION:
        raise NativeLibraryError('libocc4d.so ABI version %d != expected %d; rebuild it'
                                 % (handle.occ4d_abi_version(), ABI_VERSION))
    _lib = handle
    return _lib


def check(rc):
    """Map a status code to the exception the reference would have raised
    (AssertionError for shape/argument violations)."""
    if rc == OK:
        return
    msg = lib().occ4d_last_error().decode('utf-8', 'replace')
    if rc == EINVAL:
        raise AssertionError(msg)
    raise RuntimeError('libocc4d launch failure: ' + msg)
