"""ctypes binding of libcl4wsis_hip.so (the C ABI declared in include/cl4wsis.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CL4_LIB") or os.path.join(_HERE, "libcl4wsis_hip.so")   # CL4_LIB: kernel experiments only

ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2


class ConvDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("cin", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32),
        ("cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32),
        ("hout", C.c_int32), ("wout", C.c_int32),
        ("in_ctot", C.c_int32), ("in_coff", C.c_int32),
        ("out_ctot", C.c_int32), ("out_coff", C.c_int32),
        ("act1", C.c_int32), ("slope1", C.c_float),
        ("act2", C.c_int32), ("slope2", C.c_float),
        ("precision", C.c_int32),
    ]



class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wq", C.c_void_p), ("first", C.c_long),
                ("cout", C.c_int32), ("cin", C.c_int32), ("khw", C.c_int32), ("transposed", C.c_int32)]


class AbsJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("first", C.c_long), ("n", C.c_int32), ("eps", C.c_float)]


class LabelIO(C.Structure):
    _fields_ = [
        ("b", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("cnew", C.c_int32), ("old_classes", C.c_int32),
        ("hc", C.c_int32), ("wc", C.c_int32), ("center_ctot", C.c_int32),
        ("K", C.c_int32), ("sigma", C.c_int32), ("nms_kernel", C.c_int32), ("run_refine", C.c_int32),
        ("pseudo_thresh", C.c_float), ("refine_thresh", C.c_float), ("beta", C.c_float),
        ("seg_gt", C.c_void_p), ("prob_win", C.c_void_p), ("label_new", C.c_void_p),
        ("peak_conf", C.c_void_p), ("peak_y", C.c_void_p), ("peak_x", C.c_void_p),
        ("center_lo", C.c_void_p), ("offset_lo", C.c_void_p), ("gauss", C.c_void_p),
        ("p_center", C.c_void_p), ("p_offset", C.c_void_p), ("p_weight", C.c_void_p),
        ("r_center", C.c_void_p), ("r_offset", C.c_void_p), ("r_weight", C.c_void_p),
    ]


class P2LossIO(C.Structure):
    _fields_ = [
        ("b", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("hc", C.c_int32), ("wc", C.c_int32),
        ("cold", C.c_int32), ("cnew", C.c_int32),
        ("center", C.c_void_p), ("offset", C.c_void_p), ("old_center", C.c_void_p), ("old_offset", C.c_void_p),
        ("old_w", C.c_void_p),
        ("p_center", C.c_void_p), ("p_offset", C.c_void_p), ("p_weight", C.c_void_p),
        ("r_center", C.c_void_p), ("r_offset", C.c_void_p), ("r_weight", C.c_void_p),
        ("center_weight", C.c_float), ("offset_weight", C.c_float),
        ("losses", C.c_void_p), ("d_center", C.c_void_p), ("d_offset", C.c_void_p),
    ]


class SampleDesc(C.Structure):
    """cl4_sample_desc (include/cl4wsis.h): one decoded photo of an input batch and its crop / flip draw."""
    _fields_ = [("img_off", C.c_int64), ("lbl_off", C.c_int64), ("H", C.c_int32), ("W", C.c_int32),
                ("ci", C.c_int32), ("cj", C.c_int32), ("ch", C.c_int32), ("cw", C.c_int32), ("flip", C.c_int32),
                ("reserved", C.c_int32)]


class InsmapIO(C.Structure):
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32), ("classes", C.c_int32), ("nms_kernel", C.c_int32), ("min_mask", C.c_int32),
        ("ignore", C.c_int32), ("cap", C.c_int32),
        ("thresh", C.c_float), ("beta", C.c_float),
        ("seg_map", C.c_void_p), ("prob_win", C.c_void_p), ("center", C.c_void_p), ("offset", C.c_void_p),
        ("ins_map", C.c_void_p), ("table", C.c_void_p), ("count", C.c_void_p),
    ]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_long

# name -> argtypes; every function returns int (0 = ok). Mirrors include/cl4wsis.h.
PROTOTYPES = {}


def _decl(name, *argtypes):
    PROTOTYPES[name] = list(argtypes)


_decl("cl4_pack_conv_weight", _P, _P, _I, _I, _I, _I, _P)
_decl("cl4_pack_conv_weight_bf16x3", _P, _P, _I, _I, _I, _I, _P)
_decl("cl4_pack_conv_weight_bf16x6", _P, _P, _I, _I, _I, _I, _P)
_decl("cl4_conv2d_fwd", C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_conv2d_fwd_set_plan", C.POINTER(ConvDesc), _I, _I, _I)
_decl("cl4_conv2d_fwd_clear_plans")
_decl("cl4_conv2d_wgrad", C.POINTER(ConvDesc), _P, _P, _P, _P)
_decl("cl4_kpack", _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P)
_decl("cl4_gemm_kplanes", _P, _P, _P, _I, _I, _I, _P)
_decl("cl4_wgrad_scatter", _P, _P, _I, _I, _I, _P)
_decl("cl4_bn_fold", _P, _P, _P, _P, _F, _P, _P, _I, _P)

_decl("cl4_maxpool3x3s2", _P, _P, _L, _I, _I, _I, _I, _P)
_decl("cl4_pack_stem_weight", _P, _P, _P)
_decl("cl4_stem_fused", _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _P)
_decl("cl4_avgpool_s1", _P, _P, _L, _I, _I, _I, _I, _P)
_decl("cl4_replicate_pad", _P, _P, _L, _I, _I, _I, _I, _I, _I, _P)
_decl("cl4_global_mean", _P, _P, _L, _I, _P)
_decl("cl4_plane_sum", _P, _P, _L, _I, _P)
_decl("cl4_resize_bilinear_fwd", _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P)
_decl("cl4_resize_bilinear_bwd", _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P)
_decl("cl4_flip_w", _P, _P, _L, _I, _P)
_decl("cl4_axpy", _P, _P, _F, _L, _P)
_decl("cl4_copy_channels", _P, _P, _I, _I, _L, _I, _I, _I, _I, _P)
_decl("cl4_bn_train_stats", _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, C.c_size_t, _P)
_decl("cl4_bn_apply", _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _I, _F, _P, _F, _P)
_decl("cl4_bn_train_fwd", _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P, _F, _I, _F, _P, _F, _P, C.c_size_t, _P)
_decl("cl4_bn_train_fwd_partials", _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P, _F, _I, _F, _P, _F, _P, _I, _P)
_decl("cl4_dwconv5_fwd_stats", _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P)
_decl("cl4_bn_train_bwd", _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _I, _P, _F, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_relu_bwd", _P, _P, _P, _L, _P)
_decl("cl4_bias_grad", _P, _P, _I, _I, _I, _I, _I, _P)
_decl("cl4_dropout_mask", _P, _L, C.c_uint64, _F, _P)
_decl("cl4_dropout_mask_dev", _P, _L, _P, C.c_uint64, _F, _P)
_decl("cl4_dwconv5_fwd", _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P)
_decl("cl4_dwconv5_wgrad", _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P)

_decl("cl4_cam_smooth", _P, _I, _I, _I, _I, _P, _P, _P, _I, _F, _P, _P, C.c_size_t, _P)
_decl("cl4_conv_debug_timeline", _P)
_decl("cl4_bottleneck_fused", _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P)
_decl("cl4_bottleneck_fused_proj", _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P)
_decl("cl4_bottleneck_fused_tile", _I, _P, _P)
_decl("cl4_to_planes", _P, _P, _I, _I, _I, _I, _I, _P)
_decl("cl4_to_planes3", _P, _P, _I, _I, _I, _I, _I, _P)
_decl("cl4_from_planes", _P, _P, _I, _I, _I, _I, _I, _P)
_decl("cl4_conv2d_fwd_planes", C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_abs_eps", _P, _P, _I, _F, _P)
_decl("cl4_sign_axpy", _P, _P, _P, _I, _P)
_decl("cl4_comm_unique_id", _P)
_decl("cl4_comm_init", _I, _I, _P)
_decl("cl4_allreduce_bucket", _P, C.c_size_t, _I, _P)
_decl("cl4_allreduce_f64", _P, C.c_size_t, _P)
_decl("cl4_planes_format")
_decl("cl4_bn_apply_planes", _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _F, _I, _F, _P, _I, _F, _P)
_decl("cl4_add_act_planes", _P, _P, _P, _P, _I, _I, _I, _I, _F, _P)
_decl("cl4_pack_conv_weight_f16x3", _P, _P, _I, _I, _I, _I, _P)
_decl("cl4_to_planes_f16", _P, _P, _I, _I, _I, _I, _I, _I, _P, _P)
_decl("cl4_conv2d_fwd_planes_f16", _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_comm_destroy")
_decl("cl4_comm_world")
_decl("cl4_comm_rank")
_decl("cl4_cam_normalize", _P, _I, _I, _I, _I, _P, _P, _P, _I, _F, _P, _P, C.c_size_t, _P)
_decl("cl4_smoothing3x3", _P, _L, _I, _I, _P, _P)
_decl("cl4_peak_extract", _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_seg_pseudo_gt", _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P)
_decl("cl4_ccl", _P, _I, _I, _I, _I, _P, _P)
_decl("cl4_cc_stats", _P, _I, _I, _I, _P, _P, _P, _P)
_decl("cl4_label_generation", C.POINTER(LabelIO), _P, C.c_size_t, _P)
_decl("cl4_label_counters", _P, _I, _I, _I, _P, _P)
_decl("cl4_label_counters_accumulate", _P, _I, _I, _I, _P, _P)
_decl("cl4_p2_loss", C.POINTER(P2LossIO), _P, C.c_size_t, _P)
_decl("cl4_adam_step", _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P)
_decl("cl4_adam_step_dev", _P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P, _I, _P)
_decl("cl4_fill_zero", _P, C.c_size_t, _P)
_decl("cl4_stamp", _P, _P)
_decl("cl4_peakgen_vjp", _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_peakgen_train_loss", _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_flac_loss", _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_lloc_loss", _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_mse_loss", _P, _P, _L, _F, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_p1_seg_loss", _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _P, _P, _P, C.c_size_t, _P)
_D = C.c_double
_decl("cl4_bn_sync_stats_local", _P, _I, _I, _I, _I, _I, _P, _P, C.c_size_t, _P)
_decl("cl4_bn_sync_stats_final", _P, _D, _P, _P, _P, _P, _F, _I, _P)
_decl("cl4_bn_sync_bwd_local", _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _I, _F, _P, _F, _P, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_bn_sync_bwd_apply", _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _I, _F, _P, _F, _P, _D, _P, C.c_size_t, _P)
_decl("cl4_pack_conv_weights_f16x3_batched", _P, _I, _L, _P)
_decl("cl4_conv2d_fwd_planes_stats", _P, _P, _P, _P, _P, _P, _I, _P, C.c_size_t, _P)
_decl("cl4_bn_stats_from_partials", _P, _I, _L, _I, _P, _P, _P, _P, _F, _P)
_decl("cl4_bn_sync_stats_from_partials", _P, _I, _P, _I, _P)
_decl("cl4_abs_eps_batched", _P, _I, _L, _P)
_decl("cl4_bn_sync_bwd_local_max", _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_bn_train_bwd_act", _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _I, _F, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_bn_sync_bwd_apply_planes", _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _I, _F, _P, _D, _P, _P, C.c_size_t, _P)
_decl("cl4_zero_insert", _P, _P, _L, _I, _I, _I, _I, _I, _P)
_decl("cl4_maxpool3x3s2_bwd", _P, _P, _P, _L, _I, _I, _I, _I, _P)
_decl("cl4_add_act", _P, _P, _P, _L, _I, _F, _P)
_decl("cl4_act_bwd", _P, _P, _P, _L, _I, _F, _P)
_decl("cl4_add_per_plane", _P, _P, _L, _I, _F, _P)
_decl("cl4_dropout2d", _P, _P, _P, _L, _I, _F, _P)
_decl("cl4_rot90", _P, _P, _L, _I, _I, _I, _P)
_decl("cl4_channel_affine", _P, _P, _I, _I, _I, _P, _P, _P)
_decl("cl4_randrop_loss", _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_transpose_flip_weight", _P, _P, _I, _I, _I, _P)
_decl("cl4_random_labels", _P, _L, C.c_uint64, _I, _P)
_decl("cl4_random_labels_dev", _P, _L, _P, C.c_uint64, _I, _P)
_decl("cl4_val_seg", _P, _P, _P, _P, _I, _I, _I, _I, _I, _P)
_decl("cl4_instance_map", C.POINTER(InsmapIO), _P, C.c_size_t, _P)
_decl("cl4_semseg_hist", _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P)
_decl("cl4_sgd_step", _P, _P, _P, _L, _F, _P, _F, _F, _I, _P)
_decl("cl4_pamr", _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, C.c_size_t, _P)
_decl("cl4_ngwp_focal_bce", _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_pseudo_gtmask", _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P, _P)
_decl("cl4_softmax_channels", _P, _P, _I, _I, _I, _I, _P)
_decl("cl4_binarize", _P, _P, _P, _I, _I, _I, _I, _P)
_decl("cl4_balanced_mask_loss_ce", _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_bce_ignore_mean", _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, C.c_size_t, _P)

_decl("cl4_input_batch", _P, C.c_size_t, _P, C.c_size_t, C.POINTER(SampleDesc), _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float),
      _P, _P, _P, _P, _P, C.c_size_t, _P)
_decl("cl4_target_generation", _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, C.c_size_t, _P)

_lib = None


def load():
    """Load the shared library (once). Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). cl4wsis_amd has no CPU or PyTorch fallback path.")
    lib = C.CDLL(LIB_PATH)
    lib.cl4_abi_version.restype = C.c_int
    lib.cl4_last_error.restype = C.c_char_p
    lib.cl4_bn_workspace_bytes.restype = C.c_size_t
    lib.cl4_bn_workspace_bytes.argtypes = [C.c_int]
    lib.cl4_conv2d_fwd_planes_stats_parts.restype = C.c_int
    lib.cl4_conv2d_fwd_planes_stats_parts.argtypes = [C.POINTER(ConvDesc)]
    lib.cl4_bn_bwd_maxima_floats.restype = C.c_size_t
    lib.cl4_bn_bwd_maxima_floats.argtypes = [C.c_int]
    lib.cl4_bn_train_bwd_act_workspace_bytes.restype = C.c_size_t
    lib.cl4_bn_train_bwd_act_workspace_bytes.argtypes = [C.c_int]
    lib.cl4_conv2d_fwd_workspace_bytes.restype = C.c_size_t
    lib.cl4_conv2d_fwd_workspace_bytes.argtypes = [C.POINTER(ConvDesc)]
    for nm, at in (("cl4_peak_workspace_bytes", [C.c_int] * 5), ("cl4_label_workspace_bytes", [C.c_int] * 3),
                   ("cl4_p2_loss_workspace_bytes", [C.c_int] * 6), ("cl4_pamr_workspace_bytes", [C.c_int] * 5),
                   ("cl4_ngwp_workspace_bytes", [C.c_int] * 2), ("cl4_balanced_ce_workspace_bytes", [C.c_int] * 4),
                   ("cl4_bce_ignore_workspace_bytes", []), ("cl4_p1loss_workspace_bytes", [C.c_int] * 4), ("cl4_cam_smooth_workspace_bytes", [C.c_int] * 4),
                   ("cl4_cam_normalize_workspace_bytes", [C.c_int] * 4),
                   ("cl4_input_workspace_bytes", [C.c_int] * 5), ("cl4_target_generation_workspace_bytes", [C.c_int])):
        getattr(lib, nm).restype = C.c_size_t
        getattr(lib, nm).argtypes = at
    for nm in ("cl4_conv_debug_u_launches", "cl4_conv_debug_s_launches", "cl4_conv_debug_b_launches", "cl4_conv_debug_f_launches"):
        getattr(lib, nm).restype = C.c_long
        getattr(lib, nm).argtypes = []
    lib.cl4_stem_weight_bytes.restype = C.c_size_t
    lib.cl4_stem_weight_bytes.argtypes = []
    lib.cl4_to_planes_f16_slot_floats.restype = C.c_size_t
    lib.cl4_to_planes_f16_slot_floats.argtypes = []
    lib.cl4_kpack_bytes.restype = C.c_size_t
    lib.cl4_kpack_bytes.argtypes = [C.c_int, C.c_int]
    lib.cl4_dwconv5_stats_parts.restype = C.c_int
    lib.cl4_dwconv5_stats_parts.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.cl4_input_ksize.restype = C.c_int
    lib.cl4_input_ksize.argtypes = [C.c_int, C.c_int]
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc, name):
    if rc == 0:
        return
    lib = load()
    if rc < 0:
        raise RuntimeError(f"{name}: invalid argument: {lib.cl4_last_error().decode()}")
    raise RuntimeError(f"{name}: HIP launch failed with hipError_t={rc}")


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    check(rc, name)


def ptr(t):
    """Device pointer of a tensor (None -> NULL). The tensor must be a contiguous CUDA tensor."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("cl4wsis_amd kernels need tensors resident in HBM (device tensor expected)")
    if not t.is_contiguous():
        raise RuntimeError("cl4wsis_amd kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
