"""ctypes binding of libmhe_hip.so (the C ABI declared in include/mhe.h).

There is NO fallback: if the shared library is missing or a symbol cannot be
bound, importing the product path fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmhe_hip.so")

_p, _i, _f, _sz, _l, _d = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_long, C.c_double


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("B", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad", "dtype", "relu_in", "relu_out", "tile", "res_half")]


class WgradItem(C.Structure):
    """mhe_wgrad_item of include/mhe.h: one problem of mhe_conv_wgrad_multi_nhwc"""
    _fields_ = [("d", ConvDesc), ("x", C.c_void_p), ("gy", C.c_void_p), ("dw", C.c_void_p), ("ldw", C.c_int)]


# name -> (restype, argtypes); must list every symbol of include/mhe.h
SIGNATURES = {
    "mhe_abi_version": (_i, []),
    "mhe_last_error": (C.c_char_p, []),
    "mhe_linear_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mhe_linear_skinny_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mhe_linear_f32_bf16copy": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mhe_randn_f32": (_i, [_p, _l, _p, _f, _p]),
    "mhe_dropout": (_i, [_p, _i, _p, _l, _f, _p, _i, _p]),
    "mhe_reparam_f32": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _p]),
    "mhe_flow_packed_floats_per_net": (_sz, [_i, _i]),
    "mhe_flow_pack_net_host": (_i, [_p, _p, _p, _i, _i, _p]),
    "mhe_flow_couplings_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mhe_flow_packed_bytes_per_net_bf16": (_sz, [_i, _i]),
    "mhe_flow_pack_net_bf16_host": (_i, [_p, _p, _p, _i, _i, _p]),
    "mhe_flow_couplings_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mhe_flow_couplings_bf16_emit": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mhe_mano_table_floats": (_sz, []),
    "mhe_mano_joints_f32": (_i, [_p] * 12 + [_i, _i, _f, _f, _i, _f, _p]),
    "mhe_mano_verts_workspace_floats": (_sz, [_i]),
    "mhe_mano_verts_f32": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "mhe_mano_decode_f32": (_i, [_p] * 14 + [_i, _i, _f, _f, _i, _f, _i, _p]),
    "mhe_mano_joints_bwd_f32": (_i, [_p] * 8 + [_i, _i, _f, _f, _f, _p]),
    "mhe_sum_over_hypotheses_f32": (_i, [_p, _p, _i, _i, _i, _i, _l, _p]),
    "mhe_conv_wgrad_nhwc": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _i, _p]),
    "mhe_conv_wgrad_workspace_floats": (_sz, [C.POINTER(ConvDesc)]),
    "mhe_conv_wgrad_variant": (_i, [_p, _i, _i, _i]),
    "mhe_conv_wgrad_ws_nhwc": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _i, _p, _sz, _p]),
    "mhe_conv_wgrad_rect_workspace_floats": (_sz, [C.POINTER(ConvDesc), _i, _i]),
    "mhe_conv_wgrad_rect_nhwc": (_i, [C.POINTER(ConvDesc), _i, _i, _i, _i, _p, _p, _p, _i, _p, _sz, _p]),
    "mhe_colsum_f32": (_i, [_p, _p, _l, _i, _i, _p]),
    "mhe_colsum_workspace_floats": (_sz, [_l, _i]),
    "mhe_colsum_ws_f32": (_i, [_p, _p, _l, _i, _i, _i, _l, _p, _sz, _p]),
    "mhe_gather_f32": (_i, [_p, _p, _p, _p, _sz, _i, _p]),
    "mhe_gather_affine8_bf16": (_i, [_p, _p, _p, _p, _sz, _p]),
    "mhe_flow_mask_pad_f32": (_i, [_p, _p, _p, _l, _i, _p]),
    "mhe_flow_cond_lrelu_f32": (_i, [_p, _p, _l, _l, _i, _i, _p]),
    "mhe_flow_lrelu_bwd_f32": (_i, [_p, _p, _l, _f, _p]),
    "mhe_add_f32": (_i, [_p, _p, _p, _l, _p]),
    "mhe_flow_cond_lrelu_mixed": (_i, [_p, _i, _p, _l, _p, _p, _l, _i, _i, _p]),
    "mhe_flow_lrelu_bwd_mixed": (_i, [_p, _i, _p, _i, _p, _p, _l, _f, _p]),
    "mhe_flow_couple_bwd_f32": (_i, [_p] * 6 + [_f] + [_p] * 4 + [_l, _i, _i, _p]),
    "mhe_flow_lrelu_bwd_sum": (_i, [_p, _i, _p, _i, _p, _p, _p, _l, _p, _i, _i, _i, _f, _p]),
    "mhe_flow_mask_pad_mixed": (_i, [_p, _p, _p, _p, _l, _i, _p]),
    "mhe_flow_couple_bwd_mixed": (_i, [_p] * 6 + [_f] + [_p] * 8 + [_l, _i, _i, _p]),
    "mhe_flow_couple_accum_f32": (_i, [_p] * 5 + [_l, _i, _p]),
    "mhe_bn_mean_invstd": (_i, [_p, _p, _i, _d, _f, _p]),
    "mhe_bn_bwd_reduce_nhwc": (_i, [_p] * 5 + [_l, _i, _i, _p]),
    "mhe_bn_bwd_finalize": (_i, [_p] * 6 + [_i, _d, _p]),
    "mhe_bn_bwd_apply_nhwc": (_i, [_p] * 6 + [_l, _i, _i, _p]),
    "mhe_maxpool3x3s2_idx_nhwc": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mhe_maxpool3x3s2_bwd_nhwc": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mhe_maxpool3x3s2_idx_affine_nhwc": (_i, [_p] * 5 + [_i] * 5 + [_p]),
    "mhe_maxpool3x3s2_idx_affine_win_nhwc": (_i, [_p] * 6 + [_i] * 5 + [_p]),
    "mhe_pooled_bn_sums_nhwc": (_i, [_p] * 5 + [_l, _i, _i, _p]),
    "mhe_maxpool3x3s2_bwd_bn_nhwc": (_i, [_p] * 8 + [_i] * 5 + [_p]),
    "mhe_maxpool3x3s2_bwd_bn_apply_nhwc": (_i, [_p] * 8 + [_i] * 5 + [_p]),
    "mhe_avgpool_bwd_nhwc": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "mhe_upsample2_nhwc": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mhe_sqnorm_workspace_floats": (_sz, []),
    "mhe_sqnorm_f32": (_i, [_p, _sz, _p, _p, _p]),
    "mhe_train_tick": (_i, [_p, _p, _p]),
    "mhe_adam_step_f32": (_i, [_p] * 4 + [_sz, _p, _p] + [_f] * 6 + [_p]),
    "mhe_glow_add_image_rows_f32": (_i, [_p, _p, _l, _l, _i, _i, _i, _p]),
    "mhe_relu_copy_f32": (_i, [_p, _p, _l, _i, _p]),
    "mhe_glow_glu_residual_f32": (_i, [_p, _p, _i, _p, _l, _l, _i, _i, _i, _p]),
    "mhe_glow_coupling_f32": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _i, _p]),
    "mhe_pad64_f32": (_i, [_p, _p, _l, _i, _p]),
    "mhe_glow_coupling_inv_bwd_f32": (_i, [_p, _p, _p, _p, _f, _p, _p, _l, _i, _i, _i, _i, _p]),
    "mhe_glow_glu_bwd_f32": (_i, [_p, _p, _p, _l, _p, _p, _l, _i, _i, _i, _i, _p]),
    "mhe_relu_bwd_add_f32": (_i, [_p, _p, _p, _l, _i, _p]),
    "mhe_glow_finish_f32": (_i, [_p, _p, _p, _p, _p, _l, _i, _f, _f, _p]),
    "mhe_conv_wgrad_multi_workspace_floats": (_sz, [_p, _i]),
    "mhe_conv_wgrad_multi_nhwc": (_i, [_p, _i, _p, _sz, _p]),
    "mhe_glow_glu_bwd_sum": (_i, [_p, _p, _p, _l, _p, _p, _l, _p, _l, _i, _i, _i, _p]),
    "mhe_glow_mask_scale_sum": (_i, [_p, _p, _f, _p, _l, _i, _i, _i, _p]),
    "mhe_relu_bwd_add_mixed": (_i, [_p, _p, _p, _l, _i, _i, _p]),
    "mhe_dropout_bits": (_i, [_p, _l, _f, _p, _p]),
    "mhe_glow_layers_supported": (_i, [_i] * 6),
    "mhe_glow_layers_bf16": (_i, [_p, _p, _i] + [_p] * 13 + [_f] + [_p] * 12 + [_i] * 6 + [_l, _l, _p]),
    "mhe_glow_reverse_chain_supported": (_i, [_i] * 6),
    "mhe_glow_reverse_chain_bf16": (_i, [_p, _p, _f, _p, _p, _p, _p, _p, _i] + [_p] * 6 + [_f] + [_p] * 8 + [_i] * 6 + [_p]),
    "mhe_glow_finish_dev_f32": (_i, [_p, _p, _p, _p, _p, _l, _i, _f, _p, _i, _p]),
    "mhe_glow_affine_workspace_doubles": (_sz, [_i, _i]),
    "mhe_glow_affine_f64": (_i, [_p, _i, _i, _f] + [_p] * 7 + [_p]),
    "mhe_glow_reparam_bwd_f64": (_i, [_p, _p, _p, _i, _f, _i, _i, _p, _p, _p]),
    "mhe_mano_regress_joints_f32": (_i, [_p, _p, _p, _i, _p]),
    "mhe_elbo_reduce_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    "mhe_conv2d_nhwc": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "mhe_conv2d_masked_nhwc": (_i, [_p] * 13),
    "mhe_conv2d_f32out_nhwc": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p]),
    "mhe_conv_stat_shards": (_i, []),
    "mhe_conv1x1_residual_in_masked_nhwc": (_i, [C.POINTER(ConvDesc)] + [_p] * 15),
    "mhe_conv3x3s2_dgrad_nhwc": (_i, [_i, _i, _i, _i, _i, _i, _p, C.POINTER(C.c_void_p), _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "mhe_ho3d_geom_doubles": (_i, []),
    "mhe_ho3d_targets": (_i, [_p] * 7 + [_i] + [_p] * 18 + [_i, _p]),
    "mhe_ho3d_images": (_i, [_p] * 9 + [_i, _p]),
    "mhe_conv_tile": (_i, [C.POINTER(ConvDesc)]),
    "mhe_conv_tile_mode": (_i, [C.POINTER(ConvDesc), _i]),
    "mhe_conv1x1_residual_in_nhwc": (_i, [C.POINTER(ConvDesc)] + [_p] * 11),
    "mhe_stem_conv7x7s2": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mhe_conv1x1_stats_nhwc": (_i, [_p, _p, _p, _p, _p, _p, _p]),
    "mhe_conv_wgrad_batched_workspace_floats": (_sz, [_p, _i]),
    "mhe_conv_wgrad_batched_nhwc": (_i, [_p, _i, _p, _l, _p, _l, _p, _l, _i, _p, _sz, _p]),
    "mhe_conv2d_masked_bits_nhwc": (_i, [_p] * 14),
    "mhe_bottleneck_tail_bits_nhwc": (_i, [_p, _i] + [_p] * 15),
    "mhe_conv2d_masked_bias_nhwc": (_i, [_p, _p, _p, _i] + [_p] * 9),
    "mhe_conv3x3_halo_supported": (_i, [_i] * 5),
    "mhe_conv3x3_halo_pack_bf16": (_i, [_p, _p, _i, _i, _p]),
    "mhe_conv3x3_halo_dgrad_bn_nhwc": (_i, [_i] * 5 + [_p] * 12),
    "mhe_conv3x3_halo_nhwc": (_i, [_i] * 5 + [_p] * 5 + [_i] + [_p] * 8),
    "mhe_conv1x1_cat_bias_nhwc": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "mhe_conv3_bn_fold": (_i, [_p] * 6 + [_d] + [_p] * 4 + [_i, _p, _i] + [_p] * 2 + [_i, _i, _p]),
    "mhe_flow_reverse_chain_supported": (_i, [_i, _i, _i, _i, _i]),
    "mhe_flow_couplings_frag_supported": (_i, [_i, _i, _i, _i, _i]),
    "mhe_flow_couplings_frag_bf16": (_i, [_p, _p, _p, _i, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mhe_flow_reverse_chain_bf16": (_i, [_p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mhe_pack_transpose_bf16": (_i, [_p, _l, _p, _p, _i, _i, _p]),
    "mhe_gram_stats_words": (_sz, [_i]),
    "mhe_stat_words": (_sz, [_i]),
    "mhe_gram_stats_workspace_bytes": (_sz, [_i]),
    "mhe_conv1x1_gram_nhwc": (_i, [_p, _p, _p, _i, _p, _l, _i, _p]),
    "mhe_conv1x1_gram_store_nhwc": (_i, [_p, _p, _p, _i, _p, _p, _l, _i, _p]),
    "mhe_gram_bn_finalize": (_i, [_p] * 10 + [_i, _i, _d, _f, _f, _p, _p]),
    "mhe_bottleneck_tail_supported": (_i, [_p, _i]),
    "mhe_bottleneck_tail_nhwc": (_i, [_p, _i] + [_p] * 14),
    "mhe_stem_pool_supported": (_i, [_i, _i, _i, _i]),
    "mhe_stem_conv7x7s2_pool": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "mhe_bn_finalize": (_i, [_p] * 8 + [_i, _d, _f, _f, _p]),
    "mhe_bn_finalize_step": (_i, [_p] * 8 + [_i, _d, _f, _f, _i, _p, _p]),
    "mhe_bn_act_nhwc": (_i, [_p] * 7 + [_l, _i, _i, _i, _p]),
    "mhe_maxpool3x3s2_nhwc": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mhe_avgpool_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "mhe_bn_act_avgpool_nhwc": (_i, [_p] * 7 + [_i, _i, _i, _i, _i, _p]),
    "mhe_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "mhe_nchw_to_nhwc_pad": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mhe_rot6d_to_rotmat_f32": (_i, [_p, _p, _l, _i, _p]),
    "mhe_rot6d_to_rotmat_bwd_f32": (_i, [_p, _p, _p, _l, _p]),
    "mhe_lbs_workspace_floats": (_sz, [_i, _i, _i]),
    "mhe_lbs_pose_f32": (_i, [_p] * 7 + [_i, _i, _i, _p]),
    "mhe_lbs_skin_f32": (_i, [_p] * 6 + [_i, _i, _i, _i, _i, _f, _p]),
    "mhe_lbs_split_floats": (_sz, [_i, _i, _i]),
    "mhe_lbs_split_tables_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "mhe_lbs_skin_mfma_supported": (_i, [_i, _i, _i, _i, _i]),
    "mhe_lbs_skin_mfma_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "mhe_topk_gather_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mhe_metrics_f32": (_i, [_p] * 7 + [_i, _i, _p]),
}

_lib = None
ABI_VERSION = 4          # MHE_ABI_VERSION of include/mhe.h


class MheError(RuntimeError):
    pass


def lib():
    """Load (once) and return the bound library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MheError(
                f"{LIB_PATH} is missing: build it with `python -m mhentropy_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm ships its own HIP runtime; it has to be in the process BEFORE this library is loaded, otherwise the library
        # binds the system runtime and its kernels later find "no ROCm-capable device" next to torch's tensors
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the export is absent
            fn.restype, fn.argtypes = res, args
        if L.mhe_abi_version() != ABI_VERSION:       # struct layouts (ConvDesc) below are this version's
            raise MheError(f"{LIB_PATH} has ABI version {L.mhe_abi_version()}, these bindings are for {ABI_VERSION}: rebuild "
                           "(`python -m mhentropy_amd.build`)")
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        raise MheError(f"{what} failed ({status}): {lib().mhe_last_error().decode()}")
