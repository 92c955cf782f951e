"""ctypes binding of libpacingpseudo_hip.so (the C ABI declared in include/pacingpseudo_hip.h).

There is deliberately NO fallback: if the shared library is missing, or a call fails, the product path
raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make``)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PP_LIB_PATH') or os.path.join(_HERE, 'lib', 'libpacingpseudo_hip.so')   # override: kernel A/B tests

vp, i32, i64, f32, f64p, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_void_p, C.c_size_t


class PpLazyIn(C.Structure):
    """``pp_lazy_in`` of include/pacingpseudo_hip.h: coefficient rows of a lazy activation tensor."""
    _fields_ = [('coef', C.c_void_p), ('ld', C.c_int), ('groups', C.c_int)]


lazy_p = C.POINTER(PpLazyIn)


class PpPackItem(C.Structure):
    """``pp_pack_item`` of include/pacingpseudo_hip.h."""
    _fields_ = [('w_oihw', C.c_void_p), ('O', C.c_int), ('I', C.c_int), ('Ipad', C.c_int), ('wf16', C.c_void_p), ('wb16', C.c_void_p)]


class PpBnCoefItem(C.Structure):
    """``pp_bn_coef_item`` of include/pacingpseudo_hip.h."""
    _fields_ = [('C', C.c_int), ('groups', C.c_int), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('running_mean', C.c_void_p),
                ('running_var', C.c_void_p), ('save_mean', C.c_void_p), ('save_invstd', C.c_void_p), ('scale', C.c_void_p),
                ('shift', C.c_void_p)]


class PpWinoPackItem(C.Structure):
    """``pp_wino_pack_item`` of include/pacingpseudo_hip.h."""
    _fields_ = [('w_oihw', C.c_void_p), ('O', C.c_int), ('I', C.c_int), ('Uf16', C.c_void_p), ('Ub16', C.c_void_p)]


# name -> (restype, argtypes); mirrors include/pacingpseudo_hip.h one to one
_PROTOS = {
    'pp_version': (i32, []),
    'pp_last_error': (C.c_char_p, []),
    'pp_device_info': (i32, [C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32]),
    'pp_prof_enable': (i32, [i32]),
    'pp_prof_select': (i32, [C.c_uint64]),
    'pp_prof_reserve': (i32, [i32]),
    'pp_prof_collect': (i32, [C.POINTER(C.c_double), i32]),
    'pp_pack_image_nchw_to_nhwc': (i32, [vp, i32, i32, i32, i32, vp, i32, i32, vp]),
    'pp_pack_conv3x3_weights': (i32, [vp, i32, i32, i32, vp, vp, vp]),
    'pp_conv3x3_fwd': (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'pp_conv3x3_bwd_data': (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'pp_conv3x3_bwd_weight_workspace': (sz, [i32, i32, i32, i32, i32]),
    'pp_conv3x3_bwd_weight': (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, sz, vp]),
    'pp_pack_conv3x3_weights_f16x3': (i32, [vp, i32, i32, i32, vp, vp, vp]),
    'pp_conv3x3_fwd_f16x3': (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    'pp_conv3x3_bwd_data_f16x3': (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    'pp_conv3x3_bwd_weight_f16x3': (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, sz, vp, vp]),
    'pp_conv3x3_wino_tile': (i32, [i32, i32, i32]),
    'pp_wino_pack_weights': (i32, [vp, i32, i32, i32, vp, vp, vp]),
    'pp_conv3x3_wino_workspace': (sz, [i32, i32, i32, i32, i32, i32]),
    'pp_conv3x3_wino_vkeep_elems': (sz, [i32, i32, i32, i32, i32]),
    'pp_conv3x3_wino_fwd': (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
    'pp_conv3x3_wino_bwd_data': (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    'pp_wino_pack_weights_f16x3': (i32, [vp, i32, i32, i32, vp, vp, vp]),
    'pp_conv3x3_wino_fwd_f16x3': (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
    'pp_conv3x3_wino_bwd_data_f16x3': (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp, vp]),
    'pp_conv3x3_wino_bwd_weight_workspace': (sz, [i32, i32, i32, i32, i32, i32]),
    'pp_conv3x3_wino_bwd_weight_splits': (i32, [i32, i32, i32, i32, i32, i32]),
    'pp_conv3x3_wino_bwd_weight': (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, sz, vp]),
    'pp_conv3x3_wino_bwd_weight_f16x3': (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, sz, vp, vp]),
    'pp_bn_workspace': (sz, [i32, i32, i32]),
    'pp_bn_train_stats': (i32, [vp, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'pp_bn_eval_coeffs': (i32, [i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'pp_bn_eval_coeffs_batch': (i32, [vp, i32, f32, vp]),
    'pp_bn_lrelu_fwd': (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    'pp_bn_lrelu_fwd_pool': (i32, [vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    'pp_bn_lrelu_bwd': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32,
                              f32, vp, sz, vp]),
    'pp_bn_lrelu_bwd_amax': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32,
                                   f32, vp, sz, vp, vp]),
    'pp_bn_stats_sums': (i32, [vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    'pp_bn_train_finalize': (i32, [vp, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'pp_bn_train_finalize_lazy': (i32, [vp, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, vp]),
    'pp_lazy_materialize': (i32, [vp, i32, lazy_p, vp, i32, i32, i32, i32, vp]),
    'pp_conv3x3_bn_stats_bytes': (sz, [i32, i32, i32, i32, i32]),
    'pp_conv3x3_fwd_bn': (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, f32, i32, vp, sz,
                                C.POINTER(i32), vp]),
    'pp_conv3x3_lazy_ok': (i32, [i32, i32, i32, i32, i32, i32]),
    'pp_conv3x3_fwd_bn_lazy': (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, f32, i32, vp, sz,
                                     C.POINTER(i32), lazy_p, vp]),
    'pp_conv3x3_bwd_weight_f16x3_lazy': (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, sz, vp, lazy_p, vp]),
    'pp_conv3x3_wino_fwd_bn': (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, sz, i32, vp, vp, f32,
                                     i32, vp, sz, C.POINTER(i32), vp]),
    'pp_bn_lrelu_bwd_eval': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, f32, vp, sz, vp, vp]),
    'pp_bn_lrelu_bwd_wgrad_c1_workspace': (sz, [i32, i32, i32]),
    'pp_bn_lrelu_bwd_wgrad_c1': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, vp, i32, vp, vp, vp, i32, i32, i32,
                                       i32, f32, vp, sz, vp]),
    'pp_bn_lrelu_bwd_eval_wgrad_c1': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp, i32, i32, i32, f32,
                                            vp, sz, vp]),
    'pp_bn_lrelu_bwd_pool': (i32, [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32,
                                   f32, vp, sz, vp, vp]),
    'pp_bn_lrelu_bwd_eval_pool': (i32, [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp,
                                        sz, vp, vp]),
    'pp_bn_lrelu_bwd_sums': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, vp, vp, sz, vp]),
    'pp_bn_lrelu_bwd_apply': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32,
                                    i32, i32, f32, vp, sz, vp, vp]),
    'pp_maxpool2_fwd': (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    'pp_maxpool2_bwd': (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    'pp_bilinear_fwd': (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'pp_bilinear_bwd': (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    'pp_copy_slab': (i32, [vp, i32, vp, i32, i32, i64, i32, vp]),
    'pp_skeletonize': (i32, [vp, i32, i32, i32, vp]),
    'pp_dilate_antidiagonal': (i32, [vp, vp, i32, i32, i32, i32, vp]),
    'pp_curve_endpoints': (i32, [vp, vp, i32, i32, i32, vp]),
    'pp_aug_stats': (i32, [vp, i32, i32, i32, vp, vp, vp]),
    'pp_aug_coef': (i32, [vp, vp, vp, i32, i32, vp, vp]),
    'pp_aug_scalar_map': (i32, [vp, i32, i32, i32, vp, vp, vp]),
    'pp_aug_gamma': (i32, [vp, i32, i32, i32, vp, vp, vp]),
    'pp_aug_add_noise': (i32, [vp, i32, i32, i32, vp, vp, C.c_uint64, vp]),
    'pp_aug_warp': (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, f32, i32, i32, vp]),
    'pp_aug_spline_prefilter': (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
    'pp_aug_warp_spline': (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, f32, i32, i32, vp, vp, vp]),
    'pp_aug_elastic_field': (i32, [vp, vp, i32, i32, i32, vp, C.c_uint64, vp]),
    'pp_aug_onehot': (i32, [vp, vp, i32, i32, i32, vp]),
    'pp_aug_gaussian_blur': (i32, [vp, vp, i32, i32, i32, vp, vp]),
    'pp_aug_mix': (i32, [vp, vp, i32, i32, vp, vp]),
    'pp_aug_add_field': (i32, [vp, vp, i32, i32, i32, vp, vp]),
    'pp_stride2_gather': (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    'pp_stride2_scatter': (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    'pp_convtranspose_fwd': (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'pp_convtranspose_bwd_data': (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'pp_convtranspose_bwd_weight_workspace': (sz, [i32, i32, i32, i32, i32, i32]),
    'pp_convtranspose_bwd_weight': (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, sz, vp]),
    'pp_set_matrix_products': (i32, [i32]),
    'pp_get_matrix_products': (i32, []),
    'pp_set_wgrad_cus': (i32, [i32]),
    'pp_get_wgrad_cus': (i32, []),
    'pp_range_push': (i32, [C.c_char_p]),
    'pp_range_pop': (i32, []),
    'pp_conv1x1_nhwc_to_nchw_fwd': (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, vp]),
    'pp_conv1x1_nhwc_to_nchw_fwd_lazy': (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, lazy_p, vp]),
    'pp_conv1x1_bwd_workspace': (sz, [i32, i32, i32, i32]),
    'pp_conv1x1_nchw_to_nhwc_bwd': (i32, [vp, vp, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    'pp_conv1x1_nchw_to_nhwc_bwd_lazy': (i32, [vp, vp, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, sz, lazy_p, vp]),
    'pp_argmax_channels': (i32, [vp, i32, i32, i32, vp, vp]),
    'pp_seg_losses_workspace': (sz, [i32, i32]),
    'pp_seg_losses_fwd': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
    'pp_losses_finalize': (i32, [vp, i32, vp, vp, vp, vp]),
    'pp_seg_losses_bwd': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp]),
    'pp_aux_pce_fwd': (i32, [vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, sz, vp]),
    'pp_aux_pce_bwd': (i32, [vp, vp, i32, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'pp_memory_update_workspace': (sz, [i32, i32]),
    'pp_memory_update': (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, f32, i32, vp, sz, vp]),
    'pp_memory_update_h16': (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, f32, i32, vp, sz, vp]),
    'pp_memory_update_bf16': (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, f32, i32, vp, sz, vp]),
    'pp_memory_ce_fwd': (i32, [vp, vp, i32, i32, vp, vp]),
    'pp_memory_ce_bwd': (i32, [vp, vp, i32, i32, vp, f32, vp, i32, vp]),
    'pp_dice_counts': (i32, [vp, vp, i32, i32, i32, vp, vp]),
    'pp_hd95_workspace': (sz, [i32, i32, i32, i32]),
    'pp_hd95_surface_distances': (i32, [vp, vp, i32, i32, i32, i32, f32, f32, vp, vp, vp, sz, vp]),
    'pp_dice_loss_workspace': (sz, [i32, i32]),
    'pp_dice_loss_fwd': (i32, [vp, vp, i32, i32, i32, vp, vp, vp, sz, vp]),
    'pp_dice_loss_bwd': (i32, [vp, vp, i32, i32, i32, vp, vp, f32, vp, i32, vp]),
    'pp_adam_step': (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]),
    'pp_sgd_momentum_step': (i32, [vp, vp, vp, i64, f32, f32, f32, i32, vp]),
    'pp_channel_scale': (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    'pp_fill': (i32, [vp, i64, f32, vp]),
    'pp_scale': (i32, [vp, i64, f32, vp]),
    'pp_scale_guard': (i32, [vp, i64, f32, vp, vp]),
    'pp_adam_step_guard': (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp, vp]),
    'pp_sgd_momentum_step_guard': (i32, [vp, vp, vp, i64, f32, f32, f32, i32, vp, vp]),
    'pp_adam_step_dev': (i32, [vp, vp, vp, vp, i64, f32, vp, f32, f32, f32, f32, vp, vp, i32, vp]),
    'pp_sgd_momentum_step_dev': (i32, [vp, vp, vp, i64, f32, vp, f32, f32, vp, vp, i32, vp]),
    'pp_mfma_probe': (i32, [vp, i32, i32, C.POINTER(C.c_double), vp]),
    'pp_weighted_sum_fwd': (i32, [vp, vp, i32, vp, vp]),
    'pp_pack_conv3x3_weights_f16x3_batch': (i32, [vp, i32, vp]),
    'pp_wino_pack_weights_f16x3_batch': (i32, [vp, i32, vp]),
    'pp_weighted_sum_bwd': (i32, [vp, vp, i32, vp, vp]),
}

# 16-bit storage mode (include/pacingpseudo_hip_h16.h, generated from the sources): the entry points that read or write NHWC
# activations exist a second time with the suffix _h16 and fp16 tensors; argument lists are identical (pointers are void* here)
H16_ENTRIES = (
    'pp_conv3x3_fwd_f16x3', 'pp_conv3x3_bwd_data_f16x3', 'pp_conv3x3_fwd', 'pp_conv3x3_bwd_data', 'pp_conv3x3_bn_stats_bytes',
    'pp_conv3x3_fwd_bn', 'pp_conv3x3_lazy_ok', 'pp_conv3x3_fwd_bn_lazy', 'pp_conv3x3_bwd_weight', 'pp_conv3x3_bwd_weight_f16x3',
    'pp_conv3x3_bwd_weight_f16x3_lazy',
    'pp_conv3x3_wino_fwd_f16x3', 'pp_conv3x3_wino_fwd_bn', 'pp_conv3x3_wino_bwd_data_f16x3',
    'pp_conv3x3_wino_bwd_weight_f16x3',
    'pp_bn_workspace', 'pp_bn_train_stats', 'pp_bn_eval_coeffs', 'pp_bn_lrelu_fwd', 'pp_bn_lrelu_fwd_pool', 'pp_bn_lrelu_bwd',
    'pp_bn_lrelu_bwd_amax', 'pp_bn_stats_sums', 'pp_bn_train_finalize', 'pp_bn_train_finalize_lazy', 'pp_lazy_materialize',
    'pp_bn_lrelu_bwd_eval', 'pp_bn_lrelu_bwd_pool', 'pp_bn_lrelu_bwd_eval_pool', 'pp_bn_lrelu_bwd_sums', 'pp_bn_lrelu_bwd_apply',
    'pp_bn_lrelu_bwd_wgrad_c1_workspace', 'pp_bn_lrelu_bwd_wgrad_c1', 'pp_bn_lrelu_bwd_eval_wgrad_c1',
    'pp_pack_image_nchw_to_nhwc', 'pp_maxpool2_fwd', 'pp_maxpool2_bwd', 'pp_bilinear_fwd', 'pp_bilinear_bwd', 'pp_copy_slab',
    'pp_channel_scale',
    'pp_conv1x1_nhwc_to_nchw_fwd', 'pp_conv1x1_nhwc_to_nchw_fwd_lazy', 'pp_conv1x1_bwd_workspace', 'pp_conv1x1_nchw_to_nhwc_bwd',
    'pp_conv1x1_nchw_to_nhwc_bwd_lazy',
)
for _n in H16_ENTRIES:
    _PROTOS[_n + '_h16'] = _PROTOS[_n]
    _PROTOS[_n + '_bf16'] = _PROTOS[_n]      # round 6: the same entry points a third time, bfloat16 tensors (pacingpseudo_hip_bf16.h)
_H16_SET = frozenset(H16_ENTRIES) | {'pp_memory_update'}

EXPORTED_SYMBOLS = tuple(_PROTOS)
MIN_LIB_VERSION = 600      # include/pacingpseudo_hip.h of round 6 (pp_runtime.cpp: PP_VERSION)
PROF_KINDS = ('conv_igemm', 'conv_wgrad', 'bn', 'spatial', 'loss', 'optim', 'misc', 'wino_gemm', 'wino_wgrad',
              'wino_xform', 'conv_f16x3', 'wino_gemm_f16x3', 'wino_wgrad_f16x3', 'conv_wgrad_f16x3', 'conv_halo_f16x3')


class HipLibraryError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        self._dll = None

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise HipLibraryError(
                    f'{LIB_PATH} is missing: the pacingpseudo_amd compute path has no CPU fallback. '
                    'Build it first: python -c "import __graft_entry__ as g; g.build()"')
            dll = C.CDLL(LIB_PATH)
            # a library of another revision (PP_LIB_PATH A/B runs, scripts/build_base.sh) must say so in words, not fail with an
            # AttributeError in the middle of a step: version first, then every symbol the host side binds
            have = -1
            if hasattr(dll, 'pp_version'):
                dll.pp_version.restype = i32
                have = dll.pp_version()
            if have < MIN_LIB_VERSION:
                raise HipLibraryError(f'{LIB_PATH} reports pp_version() = {have}; this host side needs >= {MIN_LIB_VERSION} '
                                      '(an older build of the library: rebuild with `make`, or point PP_LIB_PATH at a matching one)')
            missing = [name for name in _PROTOS if not hasattr(dll, name)]
            if missing:
                raise HipLibraryError(f'{LIB_PATH} (pp_version {have}) lacks {len(missing)} entry point(s) this host side binds, e.g. '
                                      f'{", ".join(missing[:4])}: header / library mismatch, rebuild with `make`')
            for name, (res, args) in _PROTOS.items():
                fn = getattr(dll, name)
                fn.restype, fn.argtypes = res, args
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        fn = getattr(self.load(), name)
        res = _PROTOS[name][0]
        if res is not i32 or name in ('pp_version', 'pp_conv3x3_wino_tile', 'pp_conv3x3_wino_bwd_weight_splits', 'pp_conv3x3_lazy_ok', 'pp_conv3x3_lazy_ok_h16', 'pp_conv3x3_lazy_ok_bf16', 'pp_range_push', 'pp_range_pop', 'pp_get_matrix_products', 'pp_get_wgrad_cus'):      # sizes / queries / range depth: no status code
            return fn

        def checked(*a):
            rc = fn(*a)
            if rc != 0:
                msg = self.load().pp_last_error()
                raise HipLibraryError(f'{name} failed (rc={rc}): {msg.decode() if msg else "?"}')
        checked.__name__ = name
        setattr(self, name, checked)             # cache the wrapper
        return checked


lib = _Lib()


class _H16Lib:
    """The same interface as ``lib`` for a plan whose activations are stored in 16 bits (``suffix`` '_h16': IEEE fp16, '_bf16':
    bfloat16): entry points that touch activations go to their suffixed twins, everything else (weight packing, losses on fp32
    logits, optimizer ...) to the one library.  An entry point that touches activations but has no 16-bit form (strided /
    transposed convolution) raises."""
    _NO_H16 = ('pp_stride2_gather', 'pp_stride2_scatter', 'pp_convtranspose_fwd', 'pp_convtranspose_bwd_data',
               'pp_convtranspose_bwd_weight', 'pp_conv3x3_wino_fwd', 'pp_conv3x3_wino_bwd_data', 'pp_conv3x3_wino_bwd_weight')

    def __init__(self, suffix='_h16'):
        self._suffix = suffix

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        if name in self._NO_H16:
            raise HipLibraryError(f'{name} has no 16-bit storage form (include/pacingpseudo_hip{self._suffix}.h)')
        fn = getattr(lib, name + self._suffix if name in _H16_SET else name)
        setattr(self, name, fn)
        return fn


lib_h16 = _H16Lib('_h16')
lib_bf16 = _H16Lib('_bf16')
STORAGE_BYTES = {'fp32': 4, 'fp16': 2, 'bf16': 2}


def lib_for(storage):
    """The entry-point table for activations stored as `storage`: 'fp32' / 'fp16' / 'bf16' (or, as in rounds 4-5, the element
    size in bytes: 4 -> fp32, 2 -> fp16)."""
    if storage in (4, 'fp32'):
        return lib
    if storage in (2, 'fp16'):
        return lib_h16
    if storage == 'bf16':
        return lib_bf16
    raise ValueError(f'unknown activation storage {storage!r}')


class prof_range:
    """``with prof_range('backward decoder'):`` -- a named range in `rocprofv3 --marker-trace` timelines (roctx; a no-op without
    a roctx library).  PP_ROCTX=0 switches the calls off."""
    on = os.environ.get('PP_ROCTX', '1') != '0'

    def __init__(self, name: str):
        self.name = name.encode()

    def __enter__(self):
        if prof_range.on:
            lib.pp_range_push(self.name)
        return self

    def __exit__(self, *exc):
        if prof_range.on:
            lib.pp_range_pop()
        return False


def stream_ptr():
    """hipStream_t of torch's current stream (all library work is enqueued there)."""
    import torch
    return torch.cuda.current_stream().cuda_stream


def prof_collect():
    """{kernel family: dict(launches, ms, executed flops, algorithmic bytes, algorithmic flops)} accumulated since the last call."""
    n = len(PROF_KINDS)
    arr = (C.c_double * (n * 5))()
    lib.pp_prof_collect(arr, n)
    return {k: dict(launches=int(arr[i * 5]), ms=arr[i * 5 + 1], flops=arr[i * 5 + 2], bytes=arr[i * 5 + 3],
                    alg_flops=arr[i * 5 + 4]) for i, k in enumerate(PROF_KINDS)}
