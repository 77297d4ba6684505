"""ctypes binding of ``csrc/libgim_hip.so`` (the C ABI declared in ``include/gim_hip.h``).

The product path has NO fallback: if the library is missing or a tensor is not a CUDA(HIP) fp32
tensor the call raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C optimalstrategiesagainstgenerativeattacks_amd/csrc``).
"""
import ctypes
import os
from ctypes import POINTER, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GIM_LIB_PATH") or os.path.join(_HERE, "csrc", "libgim_hip.so")  # override: timing experiments


class GimConvShape(ctypes.Structure):
    _fields_ = [("N", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32),
                ("KH", c_int32), ("ups", c_int32), ("pre_slope", c_float),
                ("pool", c_int32), ("wfold", c_int32), ("res_ups", c_int32),
                ("tune_tile", c_int32), ("tune_ksplit", c_int32), ("tune_wgrad", c_int32),
                ("out_zeroed", c_int32), ("post_slope", c_float), ("prec", c_int32)]


P = c_void_p
SP = POINTER(GimConvShape)

# name -> argtypes (all return int unless noted); this table is also what tests check against the header
SIGNATURES = {
    "gim_conv2d_fwd": [P, P, P, P, P, P, SP, P],
    "gim_conv2d_dgrad": [P, P, P, P, P, SP, P],
    "gim_conv2d_dgrad_res": [P, P, P, P, P, c_float, P, SP, P],
    "gim_conv2d_wgrad_slabs": [SP],
    "gim_conv2d_wgrad": [P, P, P, P, c_int, SP, P],
    "gim_wgrad_finish": [P, P, c_int, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P],
    "gim_conv2d_fold_weights": [P, P, c_int, c_int, c_int, P],
    "gim_conv2d_fold_weights_batched": [P, P, c_int, P],
    "gim_spectral_sigma": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "gim_spectral_sigma_batched": [P, c_int, P, c_int, P, c_int, P, c_int, P],
    "gim_colsum": [P, P, P, c_int64, c_int, P],
    "gim_norm_fwd": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P],
    "gim_norm_fwd_act": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P],
    "gim_norm_bwd": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "gim_avgpool2_fwd": [P, P, c_int, c_int, c_int, c_int, P],
    "gim_avgpool2_fwd_act": [P, P, c_int, c_int, c_int, c_int, c_float, P],
    "gim_add_n": [P, P, P, P, P, c_int64, P],
    "gim_add_avgpool2_bwd": [P, P, P, c_int, c_int, c_int, c_int, P],
    "gim_avgpool2_bwd": [P, P, c_int, c_int, c_int, c_int, P],
    "gim_upsample2x_bwd": [P, P, c_float, P, c_int, c_int, c_int, c_int, P],
    "gim_maxpool_lrelu_fwd": [P, P, P, c_int, c_int, c_int, c_float, P],
    "gim_maxpool_lrelu_bwd": [P, P, P, P, c_int, c_int, c_int, c_float, P],
    "gim_bgemm": [P, P, P, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, P],
    "gim_attn_prob_fwd": [P, P, P, c_int, c_int, c_int, P],
    "gim_softmax_dim1_fwd": [P, P, c_int, c_int, c_int, P],
    "gim_softmax_dim1_bwd": [P, P, P, c_int, c_int, c_int, P],
    "gim_bgemm_grouped": [P, P, c_int, P],
    "gim_colsum_grouped": [P, P, c_int, P],
    "gim_scale_add_fwd": [P, P, P, P, c_int64, P],
    "gim_scale_add_fwd_act": [P, P, P, P, c_int64, c_float, P],
    "gim_scale_add_bwd": [P, P, P, P, P, P, c_int64, P],
    "gim_tanh_fwd": [P, P, c_int64, P],
    "gim_tanh_bwd": [P, P, P, c_int64, P],
    "gim_nchw_to_nhwc": [P, P, c_int, c_int, c_int, P],
    "gim_nhwc_to_nchw": [P, P, c_int, c_int, c_int, P],
    "gim_set_stats_fwd": [P, P, P, c_int, c_int, c_int, c_int64, c_int64, P],
    "gim_set_stats_bwd": [P, P, P, P, c_int, c_int, c_int, c_int64, c_int64, P],
    "gim_bce_logits_fwd": [P, P, c_float, c_int, P],
    "gim_bce_logits_bwd": [P, P, P, c_float, c_int, P],
    "gim_sum_dim1": [P, P, c_int, c_int, c_int, c_float, P],
    "gim_repeat_dim1": [P, P, c_int, c_int, c_int, c_float, P],
    "gim_noise_combine": [P, P, P, c_int, c_int, c_int, c_int, P],
    "gim_concat2": [P, P, P, c_int64, c_int, c_int, c_int, c_int, P],
    "gim_slice_channels": [P, P, c_int64, c_int, c_int, P],
    "gim_conv2d_wgrad_acc": [P, P, P, P, SP, P],
    "gim_wgrad_finish_batched": [P, c_int, P, c_int, P, c_int, P],
    "gim_colsum_acc": [P, P, P, c_int64, c_int, P],
    "gim_colsum2": [P, P, P, P, c_int, c_int, c_int, P],
    "gim_conv_launch_plan": [SP, c_int, P],
    "gim_conv2d_transpose_weights": [P, P, c_int, c_int, c_int, P],
    "gim_conv2d_dgrad_t": [P, P, P, P, P, SP, P],
    "gim_conv2d_xfold_weights": [P, P, c_int, c_int, c_int, c_int, P],
    "gim_conv2d_dgrad_xfold": [P, P, P, P, P, SP, c_int, P],
    "gim_episode_gather": [P, P, P, P, c_int, c_int, c_int, c_int, P],
    "gim_maxpool_gather": [P, P, P, P, c_int, c_int, c_int, c_float, P],
    "gim_softmax_dim1_bwd_dp": [P, P, P, P, c_int, c_int, c_int, P],
    "gim_set_stats_bwd_bwd": [P, P, P, P, P, P, c_int, c_int, c_int, c_int64, c_int64, P],
    "gim_lrelu_mask_mul": [P, P, c_float, P, c_int64, P],
    "gim_sqsum_rows_fwd": [P, P, c_int, c_int64, P],
    "gim_sqsum_rows_bwd": [P, P, P, c_int, c_int64, P],
    "gim_img_att_mix_fwd": [P, P, P, P, P, P, P, P, c_int64, c_int, P],
    "gim_img_att_mix_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int64, c_int, P],
    "gim_adam_step": [P, P, P, P, c_int64, P, P, c_int, c_float, c_float, c_float, c_float, P, P],
    "gim_spin": [c_int, P],
    "gim_pad_image": [P, P, c_int, c_int, c_int, c_int, c_int, c_float, P],
    "gim_conv2d_pack_rows_weights": [P, P, c_int, c_int, c_int, P],
    "gim_conv2d_pack_subpixel_weights": [P, P, c_int, c_int, c_int, P],
    "gim_depth_to_space2": [P, P, P, c_int, c_int, c_int, c_int, c_float, P],
    "gim_conv2d_fwd_rows": [P, P, P, P, P, P, SP, P],
    "gim_conv2d_wgrad_rows_acc": [P, P, P, P, SP, P],
    "gim_version": [],
}

_lib = None


def load():
    """Load (once) and return the ctypes library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libgim_hip.so not found at %s: the HIP extension is not built "
            "(run __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.gim_last_error.argtypes = []
    lib.gim_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().gim_last_error()
        raise RuntimeError("libgim_hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
