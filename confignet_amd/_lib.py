"""ctypes binding of libconfignet_hip.so (the C ABI declared in include/confignet_hip.h).

The product path has no CPU fallback: importing this module without the built library, or
calling an op without a GPU, raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CN_LIB: another build of the same sources, for A/B measurements of kernel variants; tests and the benchmark use the default)
LIB_PATH = os.environ.get("CN_LIB") or os.path.join(_HERE, "libconfignet_hip.so")


class CnSumJob(ctypes.Structure):
    """One ordered slab reduction of cn_sum_parts_grouped (include/confignet_hip.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("count", ctypes.c_longlong), ("parts", ctypes.c_int),
                ("accumulate", ctypes.c_int)]


class CnDepthJob(ctypes.Structure):
    """One shallow weight-gradient product of cn_gemm_depth_grouped (include/confignet_hip.h)."""
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("c", ctypes.c_void_p), ("m", ctypes.c_int), ("n", ctypes.c_int),
                ("k", ctypes.c_int), ("lda", ctypes.c_int), ("ldb", ctypes.c_int), ("ldc", ctypes.c_int)]


class CnRowsJob(ctypes.Structure):
    """One row-skinny dense layer of cn_gemm_rows_grouped (include/confignet_hip.h)."""
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("c", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("mask", ctypes.c_void_p)] + \
               [(n_, ctypes.c_int) for n_ in ("m", "n", "k", "lda", "ldb", "ldc", "tb", "act", "accumulate")] + [("slope", ctypes.c_float)]


class CnGanJob(ctypes.Structure):
    """One head of cn_gan_loss_grouped (include/confignet_hip.h)."""
    _fields_ = [("s", ctypes.c_void_p), ("out", ctypes.c_void_p), ("gout", ctypes.c_void_p), ("n", ctypes.c_int), ("label", ctypes.c_float)]


class CnConvGeom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "nd", "n", "in_d", "in_h", "in_w", "cin", "out_d", "out_h", "out_w", "cout",
        "k_d", "k_h", "k_w", "s_d", "s_h", "s_w", "dl_d", "dl_h", "dl_w", "p_d", "p_h", "p_w", "up")]


CN_F32, CN_BF16 = 0, 1          # dtype codes of activation tensors (include/confignet_hip.h)
CN_EUNSUPPORTED = -3


class ConfigNetHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "confignet_amd: %s is missing -- build it with `python confignet_amd/build.py` "
        "(or __graft_entry__.build()).  There is no CPU/PyTorch fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_z = ctypes.c_size_t
_G = ctypes.POINTER(CnConvGeom)

# name -> argtypes (restype is int for all but cn_last_error_string); mirrors include/confignet_hip.h
SIGNATURES = {
    "cn_version": [],
    "cn_conv_fwd": [_G, _p, _p, _p, _p, _i, _f, _p],
    "cn_conv_fwd_res": [_G, _p, _p, _p, _p, _p, _i, _f, _p],
    "cn_conv_fwd_stats": [_G, _p, _p, _p, _p, _i, _f, _p, _i, _f, _p],
    "cn_scale_columns_segments": [_p, _p, _p, _p, _i, _z, _p],
    "cn_conv_weight_tflip": [_p, _p, _i, _i, _i, _p],
    "cn_conv_dgrad": [_G, _p, _p, _p, _p],
    "cn_conv_dgrad_w": [_G, _p, _p, _p, _p],
    "cn_conv_dgrad_w_res": [_G, _p, _p, _p, _p, _p],
    "cn_bn_fold_bwd": [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "cn_conv_wgrad_thin_partials": [],
    "cn_conv_wgrad_thin": [_G, _p, _p, _p, _p, _i, _p],
    "cn_conv_wgrad": [_G, _p, _p, _p, _i, _p],
    "cn_conv_wgrad_workspace_bytes": [_G],
    "cn_conv_wgrad_ws": [_G, _p, _p, _p, _i, _p, _z, _p],
    "cn_conv_wgrad_ws_slabs": [_G, _p, _p, _p, _i, _p, _z, ctypes.POINTER(_i), _p],
    "cn_sum_parts_grouped": [_p, _i, _p],
    "cn_gemm_depth_grouped": [_p, _i, _p],
    "cn_gan_loss_grouped": [_p, _i, _i, _p],
    "cn_gemm_rows_grouped": [_p, _i, _p],
    "cn_conv_tune": [_i, _i, ctypes.c_long],
    "cn_conv_loop_select": [_i, _i, _i, _i],
    "cn_conv_fwd_dt": [_p, _p, _i, _p, _p, _p, _i, _i, _f, _p],
    "cn_conv_dgrad_dt": [_p, _p, _i, _p, _p, _i, _p],
    "cn_conv_wgrad_c3_partials": [],
    "cn_conv_wgrad_c3": [_p, _p, _p, _i, _p, _p, _i, _p],
    "cn_conv_wino_filter": [_p, _p, _i, _i, _i, _p],
    "cn_conv_fwd_wino": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _f, _p],
    "cn_conv_fwd_wino4": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _f, _p],
    "cn_conv_wino4_filter": [_p, _p, _i, _i, _i, _p],
    "cn_sumpool2": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "cn_gemm": [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _f, _p],
    "cn_nc_reduce4": [_p, _p, _i, _i, _i, _f, _i, _i, _p],
    "cn_nc_reduce": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p],
    "cn_nc_lin2": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p],
    "cn_norm_coef_fwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p],
    "cn_norm_coef_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p],
    "cn_dual_tail_coef_fwd": [_p] * 13 + [_i, _i, _i, _f, _i, _i, _p],
    "cn_dual_tail_coef_bwd": [_p] * 13 + [ctypes.POINTER(ctypes.c_void_p), _i, _i, _i, _f, _i, _i, _p],
    "cn_dual_tail_gx": [_p] * 12 + [_i, _i, _i, _f, _i, _i, _p],
    "cn_dual_tail_gx_tx": [_p] * 18 + [_i, _i, _i, _f, _i, _i, _i, _p],
    "cn_norm_apply": [_i, _i] + [_p] * 13 + [_i, _i, _i, _f, _i, _f, _i, _p],
    "cn_nc_reduce_hxt": [_p, _p, _p, _p, _i, _i, _i, _f, _i, _i, _i, _p],
    "cn_zero": [_p, ctypes.c_size_t, _p],
    "cn_set_deterministic": [_i],
    "cn_get_deterministic": [],
    "cn_det_release_stream": [_p],
    "cn_gemm_acc": [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p],
    "cn_sum_rows_into": [_p, _p, _i, _i, _i, _p],
    "cn_bn_act_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "cn_nc_reduce_dact": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _p],
    "cn_act_fwd": [_p, _p, _z, _i, _f, _i, _p],
    "cn_act_bwd": [_p, _p, _p, _z, _i, _f, _i, _p],
    "cn_act_bwd_bias": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _p],
    "cn_axpby": [_p, _p, _p, _z, _f, _f, _i, _p],
    "cn_mul": [_p, _p, _p, _z, _i, _p],
    "cn_sqdiff_sum": [_p, _p, _p, _z, _f, _i, _p],
    "cn_row_sumsq": [_p, _p, _i, _z, _p],
    "cn_row_scale": [_p, _p, _p, _i, _z, _f, _i, _p],
    "cn_row_scale_diff": [_p, _p, _p, _p, _i, ctypes.c_size_t, _f, _i, _p],
    "cn_tap_bwd": [_p, _p, _p, _p, _p, _i, ctypes.c_size_t, _f, _i, _f, _i, _p],
    "cn_maxpool2_bwd_act": [_p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p],
    "cn_masked_diff": [_p, _p, _p, _p, _z, _i, _p],
    "cn_maxpool_fwd": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "cn_maxpool_bwd": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "cn_avgpool3_same": [_p, _p, _i, _i, _i, _i, _i, _p],
    "cn_chan_affine3_fwd": [_p, _p, _z, ctypes.POINTER(_i), _f, ctypes.POINTER(_f), _p],
    "cn_chan_affine3_bwd": [_p, _p, _z, ctypes.POINTER(_i), _f, _p],
    "cn_gan_loss_fwd": [_p, _p, _i, _f, _p],
    "cn_gan_loss_bwd": [_p, _p, _p, _i, _f, _p],
    "cn_euler_matrix": [_p, _p, _i, _p],
    "cn_euler_matrix_bwd": [_p, _p, _p, _i, _p],
    "cn_rotate3d_fwd": [_p, _p, _p, _i, _i, _i, _p],
    "cn_rotate3d_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "cn_adam_step": [_p, _p, _p, _p, _p, _z, _p, _f, _f, _f, _f, _p],
    "cn_ema_step": [_p, _p, _z, _f, _p],
    "cn_gather_images_u8": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cn_to_uint8": [_p, _p, _z, _p],
    "cn_spin": [ctypes.c_ulonglong, _p],
    "cn_conv_weight_prep_bf16": [_p, _p, _p, _i, _i, _i, _p],
    "cn_conv_fwd_bf16": [_G, _p, _p, _p, _p, _i, _f, _p],
    "cn_conv_dgrad_bf16": [_G, _p, _p, _p, _p],
    "cn_conv_wgrad_bf16": [_G, _p, _p, _p, _i, _p],
    "cn_cast": [_p, _i, _p, _i, _z, _p],
    "cn_upfold_weights": [_p, _p, _p, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), _p],
    "cn_upfold_wgrad": [_p, _p, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), _i, _i, _i, _p],
    "cn_prof_enable": [_i],
    "cn_prof_reset": [],
    "cn_prof_collect": [ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
    "cn_prof_collect_by_family": [ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here == header/library mismatch
    _fn.argtypes = _args
    _fn.restype = ctypes.c_int
lib.cn_conv_wgrad_workspace_bytes.restype = ctypes.c_size_t
lib.cn_last_error_string.argtypes = []
lib.cn_last_error_string.restype = ctypes.c_char_p


def check(code, what):
    if code != 0:
        raise ConfigNetHipError("%s failed (%d): %s" % (what, code, lib.cn_last_error_string().decode()))
