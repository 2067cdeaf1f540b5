"""ctypes binding of libsemstereo_hip.so (the C ABI declared in include/semstereo_hip.h).

There is deliberately NO fallback: if the shared library is missing or a kernel
returns an error, the caller gets an exception.  PyTorch is used only for
device memory and streams; the signatures below carry raw pointers and sizes.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsemstereo_hip.so")
ABI_VERSION = 19

_P = ctypes.c_void_p
_I = ctypes.c_int

# name -> argument types (return type is always int status); mirrors include/semstereo_hip.h
_SIGNATURES = {
    "ss_groupwise_correlation_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ss_gwc_volume_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_gwc_patch_gate_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_gwc_volume_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_concat_volume_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_concat_volume_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_warp_sampled_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_warp_sampled_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_concat_sampled_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_concat_sampled_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ss_concat_sampled_presplit_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_conv3d_presplit_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_stem_left_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_stem_left_fused_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_warp_correlation_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_disparity_regression_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ss_disparity_regression_bwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ss_disparity_variance_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_softmax_regression_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_upsample_softmax_regression_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_regression_topk_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_regression_topk_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_sample_strength_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ss_topk_candidates_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ss_channel_gate_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_channel_att_logits_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_ssr_upsample_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ss_conv3d_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_conv3d_bf16s_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_conv3d_bf16s_partial_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_conv3d_gather_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_pack_conv3d_weights_bf16s": [_P, _P, _I, _I, _P],
    "ss_pack_conv3d_weights_f16s": [_P, _P, _I, _I, _P],
    "ss_conv2d_bf16s_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_conv2d_bf16s_pair_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_pack_conv2d_weights_bf16s": [_P, _P, _I, _I, _P],
    "ss_pack_conv2d_weights_f16s": [_P, _P, _I, _I, _P],
    "ss_conv3d_head_bf16s_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_conv3d_head_bf16s_cl_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_conv3d_bf16s_cl_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_pack_classifier_head_weights": [_P, _P, _P],
    "ss_conv3d_classifier_fused_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ss_regression_topk_patched_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_pack_conv3d_head_weights_bf16s": [_P, _P, _I, _P],
    "ss_pack_conv3d_head_weights_f16s": [_P, _P, _I, _P],
    "ss_conv3d_pointwise_bf16s_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_longlong, _I, _I, _P],
    "ss_pack_pointwise_weights_bf16s": [_P, _P, _I, _I, _P],
    "ss_deconv3d_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_pack_deconv3d_weights_f16s": [_P, _P, _I, _I, _P],
    "ss_deconv3d_bf16s_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_pack_deconv3d_weights_bf16s": [_P, _P, _I, _I, _I, _P],
    "ss_pack_conv3d_weights": [_P, _P, _I, _I, _I, _I, _P],
    "ss_conv3d_wgrad_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_conv3d_wgrad_bf16s_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_depthwise_patch_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_window_attention_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_window_attention_core_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_tool_copy_fwd": [_P, _P, ctypes.c_longlong, _P],
    "ss_upsample_softmax_regression_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_sample_strength_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ss_sample_strength_bwd_ws": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ss_topk_candidates_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ss_batchnorm_train_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_longlong, ctypes.c_float, _I, _P],
    "ss_batchnorm_train_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_longlong, _I, _P],
    "ss_batchnorm_train_res_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_longlong, ctypes.c_float, _I, _P],
    "ss_batchnorm_train_res_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_longlong, _I, _P],
    "ss_group_normalise_fwd": [_P, _P, _I, _I, _I, _I, _I, ctypes.c_float, _P],
    "ss_group_normalise_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_float, _P],
    "ss_batchnorm_train_fwd_rs": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_double, _I, _I, ctypes.c_longlong, ctypes.c_float, _I, _P],
    "ss_batchnorm_bwd_pg": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_longlong, _I, _P],
    "ss_batchnorm_eval_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_longlong, _I, _P],
    "ss_batchnorm_eval_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_longlong, _I, _P],
    "ss_channel_sum_fwd": [_P, _P, _I, _I, ctypes.c_longlong, _P],
    "ss_conv_k1_wgrad_fwd": [_P, _P, _P, _I, _I, _I, ctypes.c_longlong, _P],
    "ss_depthwise_patch_wgrad_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_channel_gate_bwd_logits": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ss_window_attention_core_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ss_window_attention_core_pad_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
}
EXPORTS = sorted(list(_SIGNATURES) + ["ss_abi_version", "ss_status_string", "ss_last_hip_error", "ss_ssr_param_count", "ss_reload_tuning"])

_lib = None


class SemStereoHipError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise (never fall back) if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SemStereoHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C semstereo_amd/csrc`. There is no CPU or PyTorch fallback for the HIP path.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.ss_abi_version.restype = _I
    lib.ss_status_string.restype = ctypes.c_char_p
    lib.ss_status_string.argtypes = [_I]
    lib.ss_last_hip_error.restype = ctypes.c_char_p
    lib.ss_reload_tuning.restype = _I
    lib.ss_reload_tuning.argtypes = []
    if lib.ss_abi_version() != ABI_VERSION:
        raise SemStereoHipError(f"ABI mismatch: library {lib.ss_abi_version()} != binding {ABI_VERSION}; rebuild")
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = _I
        fn.argtypes = argtypes
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an entry point on the current stream; raise on a non-zero status."""
    lib = load()
    status = getattr(lib, name)(*args, stream())
    if status != 0:
        msg = lib.ss_status_string(status).decode()
        if status == -3:
            msg += ": " + lib.ss_last_hip_error().decode()
        raise SemStereoHipError(f"{name} failed ({status}): {msg}")


def require_device(*tensors):
    """Every tensor must be fp32 on a HIP device; anything else is an error (no fallback)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise SemStereoHipError("semstereo_amd ops run on MI355X only: got a CPU tensor (no CPU fallback exists)")
        if t.dtype != torch.float32:
            raise TypeError(f"semstereo_amd ops are fp32 like the reference, got {t.dtype}")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise SemStereoHipError("all tensors of one call must live on the same device")
    return dev
