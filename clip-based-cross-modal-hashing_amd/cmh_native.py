"""ctypes binding of libcmh.so (include/cmh.h) for the Python host side.

PyTorch is plumbing here: it owns device memory and streams; every computation of the hot path
happens inside libcmh.so (hand-written HIP, gfx950).  There is NO CPU or eager-PyTorch fallback:
a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CMH_LIB") or os.path.join(_HERE, "csrc", "build", "libcmh.so")   # CMH_LIB: A/B a kernel build

ABI_VERSION = 6            # include/cmh.h CMH_VERSION: bumped whenever a struct layout or a signature changes
F32, BF16, FP8 = 0, 1, 2
ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2
TIE_REFERENCE, TIE_STABLE = 0, 1


class NativeError(RuntimeError):
    pass


class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "ln1_w", "ln1_b", "ln2_w", "ln2_b",
        "fc_w", "fc_b", "proj_w", "proj_b",
        "in_proj_cs", "out_proj_cs", "fc_cs", "proj_cs")] + [("act_scale", C.c_float * 4)]     # fp8 mode only


class GemmProblem(C.Structure):
    """include/cmh.h cmh_gemm_problem: one of the two GEMMs of a grouped launch."""
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("m_dev", C.c_void_p), ("colscale", C.c_void_p),
                ("alpha", C.c_float), ("out_scale", C.c_float)]


class VitWeights(C.Structure):
    _fields_ = [("gemm_dtype", C.c_int32), ("resolution", C.c_int32), ("patch", C.c_int32),
                ("width", C.c_int32), ("layers", C.c_int32), ("embed_dim", C.c_int32),
                ("conv1_w", C.c_void_p), ("class_embedding", C.c_void_p),
                ("positional_embedding", C.c_void_p), ("ln_pre_w", C.c_void_p), ("ln_pre_b", C.c_void_p),
                ("ln_post_w", C.c_void_p), ("ln_post_b", C.c_void_p), ("proj_t", C.c_void_p),
                ("blocks", C.POINTER(BlockWeights))]


class TextWeights(C.Structure):
    _fields_ = [("gemm_dtype", C.c_int32), ("context_length", C.c_int32), ("vocab_size", C.c_int32),
                ("width", C.c_int32), ("layers", C.c_int32), ("embed_dim", C.c_int32),
                ("token_embedding", C.c_void_p), ("positional_embedding", C.c_void_p),
                ("ln_final_w", C.c_void_p), ("ln_final_b", C.c_void_p), ("text_projection_t", C.c_void_p),
                ("blocks", C.POINTER(BlockWeights))]


class BlockGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "ln1_w", "ln1_b", "ln2_w", "ln2_b",
        "fc_w", "fc_b", "proj_w", "proj_b")]


class VitGrads(C.Structure):
    _fields_ = [("conv1_w", C.c_void_p), ("class_embedding", C.c_void_p), ("positional_embedding", C.c_void_p),
                ("ln_pre_w", C.c_void_p), ("ln_pre_b", C.c_void_p), ("ln_post_w", C.c_void_p), ("ln_post_b", C.c_void_p),
                ("proj", C.c_void_p), ("blocks", C.POINTER(BlockGrads))]


class TextGrads(C.Structure):
    _fields_ = [("token_embedding", C.c_void_p), ("positional_embedding", C.c_void_p), ("ln_final_w", C.c_void_p),
                ("ln_final_b", C.c_void_p), ("text_projection", C.c_void_p), ("blocks", C.POINTER(BlockGrads))]


class AdamTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64),
                ("lr", C.c_float), ("weight_decay", C.c_float), ("max_grad_norm", C.c_float), ("p_bf16", C.c_void_p)]


class Taps(C.Structure):
    _fields_ = [("ptrs", C.POINTER(C.c_void_p)), ("count", C.c_int32)]


_lib = None
_lock = threading.Lock()

_i32, _i64, _f, _p, _sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); exactly the symbols include/cmh.h declares
SIGNATURES = {
    "cmh_last_error": (C.c_char_p, []),
    "cmh_version": (C.c_int, []),
    "cmh_vit_workspace_bytes": (_sz, [C.POINTER(VitWeights), _i32]),
    "cmh_text_workspace_bytes": (_sz, [C.POINTER(TextWeights), _i32, _i32]),
    "cmh_vit_encode": (C.c_int, [C.POINTER(VitWeights), _p, _i32, _p, _p, _sz, C.POINTER(Taps), _p]),
    "cmh_text_encode": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, _p, _sz, C.POINTER(Taps), _p]),
    "cmh_text_encode_packed": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, _p, _sz, _p]),
    "cmh_vit_calibrate_fp8": (C.c_int, [C.POINTER(VitWeights), _p, _i32, _p, _p, _p, _sz, _p]),
    "cmh_text_calibrate_fp8": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, _p, _sz, _p]),
    "cmh_linear_gemm": (C.c_int, [_i32, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "cmh_layernorm": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "cmh_attention": (C.c_int, [_i32, _p, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "cmh_gemm_tuning": (C.c_int, [_i32, _i32]),
    "cmh_set_pooled_tail": (C.c_int, [_i32]),
    "cmh_set_gemm_rows": (C.c_int, [_i32]),
    "cmh_set_gemm_grouped": (C.c_int, [_i32]),
    "cmh_set_gemm_lc": (C.c_int, [_i32]),
    "cmh_set_grad_stream16": (C.c_int, [_i32]),
    "cmh_set_text_token_packing": (C.c_int, [_i32]),
    "cmh_linear_gemm_grouped": (C.c_int, [_i32, C.POINTER(GemmProblem), C.POINTER(GemmProblem), _i32, _p]),
    "cmh_clip_encode_pair": (C.c_int, [C.POINTER(VitWeights), _p, C.POINTER(TextWeights), _p, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _p, _sz, _p]),
    "cmh_clip_encode_pair2": (C.c_int, [C.POINTER(VitWeights), _p, _i32, _p, _i32, C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, _p, _p, _sz, _p, _sz, _p]),
    "cmh_msl_workspace_bytes": (_sz, [_i32]),
    "cmh_msl_loss": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "cmh_msl_loss_backward": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "cmh_spl_workspace_bytes": (_sz, [_i32]),
    "cmh_spl_loss": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _f, _f, _p, _p, _sz, _p]),
    "cmh_spl_loss_backward": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _f, _f, _p, _p, _p, _p, _sz, _p]),
    "cmh_qmi_workspace_bytes": (_sz, [_i32]),
    "cmh_qmi_loss": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _f, _p, _p, _p, _sz, _p]),
    "cmh_qmi_loss_backward": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _f, _p, _p, _p, _p, _p, _sz, _p]),
    "cmh_fp8_quantize_weight": (C.c_int, [_p, _p, _p, _i32, _i32, _p]),
    "cmh_fp8_quantize": (C.c_int, [_p, _i32, _p, _i64, _f, _p]),
    "cmh_fp8_dequantize": (C.c_int, [_p, _p, _i64, _f, _p]),
    "cmh_amax": (C.c_int, [_p, _i32, _i64, _p, _p]),
    "cmh_linear_gemm_fp8": (C.c_int, [_p, _p, _p, _f, _p, _p, _p, _f, _i32, _i32, _i32, _i32, _p]),
    "cmh_prof_gemm_begin": (C.c_int, [_i32]),
    "cmh_prof_gemm_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "cmh_prof_gemm_by_kernel": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "cmh_cast_f32_to_bf16": (C.c_int, [_p, _p, _i64, _p]),
    "cmh_linear_act": (C.c_int, [_p, _p, _p, _p, _f, _i32, _p, _i32, _i32, _i32, _p]),
    "cmh_pair_softmax": (C.c_int, [_p, _p, _i32, _i32, _p]),
    "cmh_sign_codes": (C.c_int, [_p, _p, _i64, _p]),
    "cmh_pair_argmax_codes": (C.c_int, [_p, _p, _i32, _i32, _p]),
    "cmh_pack_codes": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p]),
    "cmh_unpack_codes": (C.c_int, [_p, _p, _i64, _i32, _p, _p]),
    "cmh_map_mean": (C.c_int, [_p, _i32, _p, _p]),
    "cmh_pack_labels": (C.c_int, [_p, _i64, _i32, _p, _p, _p]),
    "cmh_hamming_dist": (C.c_int, [_p, _p, _p, _p, _i32, _i64, _i32, _p, _p]),
    "cmh_calc_neighbor": (C.c_int, [_p, _p, _i32, _i32, _i32, _p, _p]),
    "cmh_map_workspace_bytes": (_sz, [_i32, _i64, _i32, _i32]),
    "cmh_hamming_map": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i64, _i32, _i32, _i64, _i32, _i32, _p, _p, _p,
                                  _p, _sz, _p]),
    "cmh_loss_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "cmh_dsph_hyp_loss": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _f, _f, _p, _p, _sz, _p]),
    "cmh_dchmt_loss": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _f, _f, _p, _p, _sz, _p]),
    "cmh_batchnorm1d_train": (C.c_int, [_p, _p, _p, _f, _p, _i32, _i32, _p]),
    "cmh_twdh_targets": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "cmh_twdh_loss": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _p, _sz, _p]),
    "cmh_dnph_loss": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f, _f, _p, _p, _sz, _p]),
    "cmh_vit_encode_tokens": (C.c_int, [C.POINTER(VitWeights), _p, _i32, _p, _p, _sz, _p]),
    "cmh_text_encode_tokens": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "cmh_text_encode_tokens_packed": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "cmh_blocks_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "cmh_transformer_blocks": (C.c_int, [C.POINTER(BlockWeights), _i32, _i32, _p, _i32, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "cmh_mith_lta": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cmh_add_positional": (C.c_int, [_p, _p, _i32, _i32, _i32, _p]),
    "cmh_bitwise_hash": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "cmh_l2_normalize_rows": (C.c_int, [_p, _p, _i32, _i32, _p]),
    "cmh_mith_mix": (C.c_int, [_p, _p, _p, _p, _f, _p, _p, _p, _i64, _p]),
    "cmh_sq_diff_sum": (C.c_int, [_p, _p, _i64, _p, _p, _sz, _p]),
    "cmh_mith_bayesian_loss": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "cmh_info_nce": (C.c_int, [_p, _p, _i32, _i32, _i32, _f, _p, _p, _sz, _p]),
    "cmh_transpose": (C.c_int, [_p, _i32, _p, _i32, _i32, _i32, _p]),
    "cmh_colsum_workspace_bytes": (_sz, [_i32, _i32]),
    "cmh_colsum": (C.c_int, [_p, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "cmh_layernorm_backward_workspace_bytes": (_sz, [_i32, _i32]),
    "cmh_layernorm_backward": (C.c_int, [_p, _i32, _p, _i32, _p, _i32, _i32, _p, _i32, _p, _p, _p, _sz, _p]),
    "cmh_quick_gelu": (C.c_int, [_p, _p, _i64, _i32, _p]),
    "cmh_attention_backward": (C.c_int, [_i32, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "cmh_linear_wgrad_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "cmh_linear_wgrad": (C.c_int, [_i32, _p, _i32, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _sz, _p]),
    "cmh_linear_act_backward": (C.c_int, [_p, _p, _p, _p, _p, _f, _i32, _p, _p, _p, _i32, _i32, _i32, _p, _sz, _p]),
    "cmh_head_backward_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "cmh_pair_softmax_backward": (C.c_int, [_p, _p, _p, _i32, _i32, _p]),
    "cmh_dchmt_loss_backward": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _f, _f, _p, _p, _p, _p, _sz, _p]),
    "cmh_dsph_hyp_loss_backward": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _f, _f, _p, _p, _p, _p, _p, _sz, _p]),
    "cmh_dnph_backward_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "cmh_dnph_loss_backward": (C.c_int, [_p] * 8 + [_i32, _i32, _i32, _f, _f, _p] + [_p] * 5 + [_p, _sz, _p]),
    "cmh_image_preprocess_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "cmh_image_preprocess": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       _p, _p, _p, _sz, _p]),
    "cmh_vit_forward_train_tokens": (C.c_int, [C.POINTER(VitWeights), _p, _i32, _p, _p, _sz, _p]),
    "cmh_vit_backward_tokens": (C.c_int, [C.POINTER(VitWeights), _i32, _p, C.POINTER(VitGrads), _p, _sz, _p]),
    "cmh_text_forward_train_tokens": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "cmh_text_forward_train_tokens_packed": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "cmh_text_backward_tokens": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, C.POINTER(TextGrads), _p, _sz, _p]),
    "cmh_blocks_train_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "cmh_blocks_forward_train": (C.c_int, [C.POINTER(BlockWeights), _i32, _i32, _p, _p, _i32, _i32, _i32, _p, _sz, _p]),
    "cmh_blocks_backward": (C.c_int, [C.POINTER(BlockWeights), C.POINTER(BlockGrads), _i32, _i32, _p, _p, _i32, _i32, _i32, _p, _sz, _p]),
    "cmh_mith_lta_backward": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cmh_gelu": (C.c_int, [_p, _p, C.c_int64, _p]),
    "cmh_gelu_backward": (C.c_int, [_p, _p, _p, C.c_int64, _p]),
    "cmh_l2_normalize_backward": (C.c_int, [_p, _p, _p, _i32, _i32, _p]),
    "cmh_bitwise_hash_backward": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "cmh_mith_bayesian_backward_workspace_bytes": (_sz, [_i32, _i32]),
    "cmh_info_nce_workspace_bytes": (_sz, [_i32, _i32]),
    "cmh_mith_bayesian_loss_backward": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _sz, _p]),
    "cmh_info_nce_backward": (C.c_int, [_p, _p, _i32, _i32, _i32, _f, _p, _p, _p, _p, _sz, _p]),
    "cmh_sq_diff_sum_backward": (C.c_int, [_p, _p, C.c_int64, _p, _p, _p, _p]),
    "cmh_batchnorm1d_update_running": (C.c_int, [_p, _f, _p, _p, _i32, _i32, _p]),
    "cmh_batchnorm1d_backward": (C.c_int, [_p, _p, _f, _p, _p, _p, _p, _i32, _i32, _p]),
    "cmh_twdh_loss_backward": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p]),
    "cmh_image_normalize": (C.c_int, [_p, _p, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float), _p, _p]),
    "cmh_bpe_create": (C.c_int, [C.c_char_p, _sz, C.POINTER(_p)]),
    "cmh_bpe_destroy": (None, [_p]),
    "cmh_bpe_vocab_size": (_i32, [_p]),
    "cmh_bpe_encode_captions": (C.c_int, [_p, C.c_char_p, _p, _i32, _i32, _p, _p, _i32]),
    "cmh_vit_train_bytes": (_sz, [C.POINTER(VitWeights), _i32]),
    "cmh_vit_forward_train": (C.c_int, [C.POINTER(VitWeights), _p, _i32, _p, _p, _sz, _p]),
    "cmh_vit_backward": (C.c_int, [C.POINTER(VitWeights), _i32, _p, C.POINTER(VitGrads), _p, _sz, _p]),
    "cmh_vit_backward_part": (C.c_int, [C.POINTER(VitWeights), _i32, _p, C.POINTER(VitGrads), _p, _sz, _i32, _i32, _p]),
    "cmh_text_train_bytes": (_sz, [C.POINTER(TextWeights), _i32, _i32]),
    "cmh_text_forward_train": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, _p, _sz, _p]),
    "cmh_text_backward": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, C.POINTER(TextGrads), _p, _sz, _p]),
    "cmh_text_backward_part": (C.c_int, [C.POINTER(TextWeights), _p, _i32, _i32, _p, _p, C.POINTER(TextGrads), _p, _sz, _i32, _i32, _p]),
    "cmh_bert_adam_workspace_bytes": (_sz, [_i32, _i64]),
    "cmh_bert_adam_step": (C.c_int, [C.POINTER(AdamTensor), _i32, C.c_double, C.c_double, C.c_double, _p, _sz, _p]),
}


def lib():
    """Load libcmh.so (built by `__graft_entry__.build()` / `make -C csrc`).  Fails loudly."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NativeError(
                    f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or `make -C "
                    f"{os.path.dirname(os.path.dirname(LIB_PATH))}`); there is no CPU fallback for the hot path")
            l = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(l, name)          # AttributeError if the ABI lost a symbol
                fn.restype, fn.argtypes = res, args
            if l.cmh_version() != ABI_VERSION:  # a stale or diagnostic build with other struct layouts would read garbage pointers
                raise NativeError(f"{LIB_PATH}: cmh_version() = {l.cmh_version()}, this binding was written for {ABI_VERSION}; rebuild it")
            _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise NativeError(f"{what} failed (code {rc}): {lib().cmh_last_error().decode(errors='replace')}")


# ------------------------------------------------------------------------------------------ helpers
def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NativeError("libcmh runs on the GPU only (tensor on %s); there is no CPU fallback" % t.device)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_ws_cache: dict = {}


def workspace(nbytes: int, device, tag: str = "") -> torch.Tensor:
    """Per-(device, CURRENT STREAM, tag) grow-only scratch buffer (uint8, 256-byte aligned by the caching allocator).  One buffer per
    stream: the towers, the two branches of MITH's HashingModel and consecutive batches of the evaluation loops run on different
    streams at the same time, and a scratch buffer shared between them would be a data race (round-4 advisor finding)."""
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream) if torch.cuda.is_available() else 0, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def cast_bf16(src: torch.Tensor) -> torch.Tensor:
    """f32 -> bf16 copy through the library's RNE cast kernel."""
    src = f32c(src)
    require_gpu(src)
    dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    check(lib().cmh_cast_f32_to_bf16(ptr(src), ptr(dst), src.numel(), stream_ptr(src.device)), "cmh_cast_f32_to_bf16")
    return dst


def _kind(t: torch.Tensor) -> int:
    return {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[t.dtype]


def fp8_quantize_weight(w: torch.Tensor):
    """w f32 [N,K] -> (e4m3 bytes [N,K] uint8, colscale f32 [N]) with w ~ fp8 * colscale[:, None]."""
    w = f32c(w)
    require_gpu(w)
    Nn, K = w.shape
    q = torch.empty(Nn, K, dtype=torch.uint8, device=w.device)
    cs = torch.empty(Nn, dtype=torch.float32, device=w.device)
    check(lib().cmh_fp8_quantize_weight(ptr(w), ptr(q), ptr(cs), Nn, K, stream_ptr(w.device)), "cmh_fp8_quantize_weight")
    return q, cs


def fp8_quantize(x: torch.Tensor, scale: float) -> torch.Tensor:
    """x (f32 / bf16 / f16) -> e4m3 bytes of clamp(x / scale, +-448), same shape, uint8."""
    require_gpu(x)
    x = x.contiguous()
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(lib().cmh_fp8_quantize(ptr(x), _kind(x), ptr(q), x.numel(), float(scale), stream_ptr(x.device)), "cmh_fp8_quantize")
    return q


def fp8_dequantize(q: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    require_gpu(q)
    q = q.contiguous()
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    check(lib().cmh_fp8_dequantize(ptr(q), ptr(out), q.numel(), float(scale), stream_ptr(q.device)), "cmh_fp8_dequantize")
    return out


def amax(x: torch.Tensor, into: torch.Tensor | None = None) -> torch.Tensor:
    """max |x| as a device scalar (accumulated into `into` when given)."""
    require_gpu(x)
    x = x.contiguous()
    out = torch.zeros(1, dtype=torch.float32, device=x.device) if into is None else into
    check(lib().cmh_amax(ptr(x), _kind(x), x.numel(), ptr(out), stream_ptr(x.device)), "cmh_amax")
    return out


EPI_OUT_FP8 = 4096


def linear_gemm_fp8(x8, w8, colscale, alpha, bias=None, residual=None, quickgelu=False, out="f32", out_scale=1.0):
    """epi(alpha * colscale[n] * (x8 @ w8.T)) on e4m3 operands (uint8 tensors); out in {"f32", "bf16", "f16", "fp8"}."""
    require_gpu(x8, w8, colscale, bias, residual)
    if x8.dtype != torch.uint8 or w8.dtype != torch.uint8:
        raise NativeError("linear_gemm_fp8: operands are e4m3 bytes (uint8 tensors)")
    x8, w8 = x8.contiguous(), w8.contiguous()
    M, K = x8.shape
    Nn = w8.shape[0]
    if w8.shape[1] != K or colscale.numel() != Nn or (bias is not None and bias.numel() != Nn) or \
            (residual is not None and tuple(residual.shape) != (M, Nn)):
        raise NativeError(f"linear_gemm_fp8: x {tuple(x8.shape)}, w {tuple(w8.shape)}, colscale / bias / residual shapes do not fit together")
    odt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "fp8": torch.uint8}[out]
    o = torch.empty(M, Nn, dtype=odt, device=x8.device)
    res_f16 = residual is not None and residual.dtype == torch.float16
    if residual is not None:
        residual = residual.contiguous() if res_f16 else f32c(residual)
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_QUICKGELU if quickgelu else 0) | (EPI_RESIDUAL if residual is not None else 0) | \
          (EPI_RES_F16 if res_f16 else 0) | {"f32": 0, "bf16": EPI_OUT_BF16, "f16": EPI_OUT_F16, "fp8": EPI_OUT_FP8}[out]
    check(lib().cmh_linear_gemm_fp8(ptr(x8), ptr(w8), ptr(f32c(colscale)), float(alpha), ptr(None if bias is None else f32c(bias)),
                                    ptr(residual), ptr(o), float(out_scale), M, Nn, K, epi, stream_ptr(x8.device)), "cmh_linear_gemm_fp8")
    return o


def set_gemm_grouped(on: int = -1):
    """Layer i of both towers as ONE grouped GEMM launch (csrc/gemm_wide.hip, GRP): 1 on (default), 0 = two plain launches, -1 = environment
    (CMH_GEMM_GROUPED=0 is off).  Results never depend on it."""
    check(lib().cmh_set_gemm_grouped(int(on)), "cmh_set_gemm_grouped")


def linear_gemm_grouped(problems, quickgelu=False, out="f32", m_dev=(None, None)):
    """Two GEMMs of one kind as one launch (include/cmh.h: cmh_linear_gemm_grouped).  problems = two dicts with x, w and optionally
    bias, residual, and - e4m3 operands (uint8 tensors) - colscale, alpha, out_scale; out in {"f32", "bf16", "f16", "fp8"};
    m_dev: per problem an int32 device tensor holding the real row count (or None).  Returns the two outputs."""
    assert len(problems) == 2
    kinds = {("u8" if p["x"].dtype == torch.uint8 else ("bf16" if p["x"].dtype == torch.bfloat16 else "f32")) for p in problems}
    if len(kinds) != 1:
        raise NativeError("linear_gemm_grouped: both problems must share the operand type")
    kind = kinds.pop()
    dt = {"f32": F32, "bf16": BF16, "u8": FP8}[kind]
    odt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "fp8": torch.uint8}[out]
    has_bias = problems[0].get("bias") is not None
    has_res = problems[0].get("residual") is not None
    res_f16 = has_res and problems[0]["residual"].dtype == torch.float16
    epi = (EPI_BIAS if has_bias else 0) | (EPI_QUICKGELU if quickgelu else 0) | (EPI_RESIDUAL if has_res else 0) | \
          (EPI_RES_F16 if res_f16 else 0) | {"f32": 0, "bf16": EPI_OUT_BF16, "f16": EPI_OUT_F16, "fp8": EPI_OUT_FP8}[out]
    structs, outs, keep = [], [], []
    for p, md in zip(problems, m_dev):
        x, w = p["x"].contiguous(), p["w"].contiguous()
        require_gpu(x, w, p.get("bias"), p.get("residual"), p.get("colscale"), md)
        M, K = x.shape
        Nn = w.shape[0]
        if w.shape[1] != K or (p.get("bias") is not None) != has_bias or (p.get("residual") is not None) != has_res or \
                (has_bias and p["bias"].numel() != Nn) or (has_res and tuple(p["residual"].shape) != (M, Nn)) or \
                (kind == "u8" and (p.get("colscale") is None or p["colscale"].numel() != Nn)):
            raise NativeError(f"linear_gemm_grouped: x {tuple(x.shape)}, w {tuple(w.shape)}: operand shapes / kinds do not fit together")
        o = torch.empty(M, Nn, dtype=odt, device=x.device)
        bias = f32c(p["bias"]) if has_bias else None
        res = None
        if has_res:
            res = p["residual"].contiguous() if res_f16 else f32c(p["residual"])
        cs = f32c(p["colscale"]) if kind == "u8" else None
        g = GemmProblem(ptr(x), ptr(w), ptr(bias), ptr(res), ptr(o), M, Nn, K, ptr(md), ptr(cs), float(p.get("alpha", 1.0)),
                        float(p.get("out_scale", 1.0)))
        structs.append(g)
        outs.append(o)
        keep += [x, w, bias, res, cs]
    check(lib().cmh_linear_gemm_grouped(dt, C.byref(structs[0]), C.byref(structs[1]), epi, stream_ptr(outs[0].device)),
          "cmh_linear_gemm_grouped")
    return outs


def set_pooled_tail(on: bool):
    """Carry only the pooled rows through the last block of encode_image / encode_text (default on; include/cmh.h)."""
    check(lib().cmh_set_pooled_tail(1 if on else 0), "cmh_set_pooled_tail")


def set_gemm_rows(on: int = -1):
    """Few-row GEMMs (M <= 512) on 64 x 64 tiles (csrc/gemm_rows.hip): 1 on (default), 0 = the wide kernel takes them, -1 = environment."""
    check(lib().cmh_set_gemm_rows(int(on)), "cmh_set_gemm_rows")


def set_text_token_packing(on: int):
    """the all-token text trunk (MITH) skips the positions behind a caption's last unpadded token (1, the default); 0: computes every
    position; -1 = CMH_TEXT_PACK_TOKENS"""
    check(lib().cmh_set_text_token_packing(int(on)), "cmh_set_text_token_packing")


def set_grad_stream16(on: int):
    """bf16 training mode: carry the towers' residual-gradient stream as bf16 (1, the default), as f32 (0); -1 = CMH_GRAD_STREAM16"""
    check(lib().cmh_set_grad_stream16(int(on)), "cmh_set_grad_stream16")


def set_gemm_lc(mode: int):
    """Loader / consumer GEMM kernel (csrc/gemm_lc.hip): 0 never, 1 every eligible launch, 2 all but QuickGELU launches, 3 cost model."""
    check(lib().cmh_set_gemm_lc(int(mode)), "cmh_set_gemm_lc")


def gemm_tuning(tile_rows: int = -1, order_group: int = -1):
    """Pin the wide GEMM's tile height (96 / 128 / 160) and tile-order group (0 = n-fastest); -1 = automatic."""
    check(lib().cmh_gemm_tuning(int(tile_rows), int(order_group)), "cmh_gemm_tuning")


def prof_gemm_begin(max_launches: int):
    check(lib().cmh_prof_gemm_begin(int(max_launches)), "cmh_prof_gemm_begin")


def prof_gemm_end():
    """-> (sum of GEMM launch durations [ms], sum of algorithmic FLOPs, number of launches)"""
    ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
    check(lib().cmh_prof_gemm_end(C.byref(ms), C.byref(fl), C.byref(n)), "cmh_prof_gemm_end")
    return ms.value, fl.value, n.value


def prof_gemm_by_kernel():
    """after prof_gemm_end(): {kernel: (ms, flops, launches)} of the timed launches, split by the kernel that ran them"""
    ms, fl, n = (C.c_double * 3)(), (C.c_double * 3)(), (C.c_int64 * 3)()
    check(lib().cmh_prof_gemm_by_kernel(ms, fl, n), "cmh_prof_gemm_by_kernel")
    return {k: (ms[i], fl[i], n[i]) for i, k in enumerate(("gemm_wide_kernel", "gemm_rows_kernel", "fallback"))}


# ------------------------------------------------------------------------------------------ tower blocks
EPI_BIAS, EPI_QUICKGELU, EPI_RESIDUAL, EPI_OUT_BF16 = 1, 2, 4, 8
EPI_RES_F16, EPI_OUT_F16 = 64, 128


def linear_gemm(x, w, bias=None, residual=None, quickgelu=False, out_bf16=False, out_f16=False):
    """out = epi(x @ w.T); x,w both f32 or both bf16 (dtype picks the MFMA path).  A float16 `residual` / `out_f16` is the
    bf16 mode's fp16 residual stream (CMH_EPI_RES_F16 / CMH_EPI_OUT_F16, N % 256 == 0)."""
    require_gpu(x, w, bias, residual)
    dt = BF16 if x.dtype == torch.bfloat16 else F32
    if (w.dtype == torch.bfloat16) != (dt == BF16):
        raise NativeError("linear_gemm: x and w must share a dtype")
    if out_bf16 and out_f16:
        raise NativeError("linear_gemm: out_bf16 and out_f16 exclude each other")
    x, w = x.contiguous(), w.contiguous()
    M, K = x.shape
    Nn = w.shape[0]
    if w.dim() != 2 or w.shape[1] != K or (bias is not None and bias.numel() != Nn) or \
            (residual is not None and tuple(residual.shape) != (M, Nn)):
        raise NativeError(f"linear_gemm: x {tuple(x.shape)}, w {tuple(w.shape)}, bias / residual shapes do not fit together")
    odt = torch.bfloat16 if out_bf16 else (torch.float16 if out_f16 else torch.float32)
    out = torch.empty(M, Nn, dtype=odt, device=x.device)
    res_f16 = residual is not None and residual.dtype == torch.float16
    if residual is not None:
        residual = residual.contiguous() if res_f16 else f32c(residual)
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_QUICKGELU if quickgelu else 0) | \
          (EPI_RESIDUAL if residual is not None else 0) | (EPI_OUT_BF16 if out_bf16 else 0) | \
          (EPI_RES_F16 if res_f16 else 0) | (EPI_OUT_F16 if out_f16 else 0)
    check(lib().cmh_linear_gemm(dt, ptr(x), ptr(w), ptr(None if bias is None else f32c(bias)), ptr(residual), ptr(out),
                                M, Nn, K, epi, stream_ptr(x.device)), "cmh_linear_gemm")
    return out


def layernorm(x, w, b, out_bf16=False):
    x, w, b = f32c(x), f32c(w), f32c(b)
    require_gpu(x, w, b)
    M, d = x.shape
    if w.numel() != d or b.numel() != d:
        raise NativeError(f"layernorm: rows of {d} elements, weight {tuple(w.shape)}, bias {tuple(b.shape)}")
    out = torch.empty(M, d, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    check(lib().cmh_layernorm(ptr(x), ptr(w), ptr(b), ptr(out), BF16 if out_bf16 else F32, M, d, stream_ptr(x.device)),
          "cmh_layernorm")
    return out


def attention(qkv, B, T, causal, key_padding_mask=None):
    require_gpu(qkv, key_padding_mask)
    qkv = qkv.contiguous()
    dt = BF16 if qkv.dtype == torch.bfloat16 else F32
    d = qkv.shape[1] // 3
    o = torch.empty(B * T, d, dtype=qkv.dtype, device=qkv.device)
    kpm = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
    check(lib().cmh_attention(dt, ptr(qkv), ptr(o), B, T, d, int(bool(causal)), ptr(kpm), stream_ptr(qkv.device)),
          "cmh_attention")
    return o


# ------------------------------------------------------------------------------------------ heads
def linear_act(x, w, b, act=ACT_NONE, drop_mask=None, p=0.2):
    x, w = f32c(x), f32c(w)
    b = None if b is None else f32c(b)
    require_gpu(x, w, b, drop_mask)
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise NativeError(f"linear_act: x [{M},{K}] vs w {tuple(w.shape)}")
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    dm = None if drop_mask is None else f32c(drop_mask)
    check(lib().cmh_linear_act(ptr(x), ptr(w), ptr(b), ptr(dm), 1.0 / (1.0 - p), act, ptr(y), M, N, K,
                               stream_ptr(x.device)), "cmh_linear_act")
    return y


def pair_softmax(z):
    z = f32c(z)
    require_gpu(z)
    M, K2 = z.shape
    out = torch.empty_like(z)
    check(lib().cmh_pair_softmax(ptr(z), ptr(out), M, K2 // 2, stream_ptr(z.device)), "cmh_pair_softmax")
    return out


def sign_codes(h):
    h = f32c(h)
    require_gpu(h)
    out = torch.empty_like(h)
    check(lib().cmh_sign_codes(ptr(h), ptr(out), h.numel(), stream_ptr(h.device)), "cmh_sign_codes")
    return out


def pair_argmax_codes(p):
    p = f32c(p)
    require_gpu(p)
    M, K2 = p.shape
    out = torch.empty(M, K2 // 2, dtype=torch.float32, device=p.device)
    check(lib().cmh_pair_argmax_codes(ptr(p), ptr(out), M, K2 // 2, stream_ptr(p.device)), "cmh_pair_argmax_codes")
    return out


# ------------------------------------------------------------------------------------------ hamming / mAP
def pack_codes(codes, validate=True):
    """f32 codes in {-1,0,+1} [n, K] -> (sign_plane, nz_plane) int32 [n, ceil(K/32)].
    validate=True reads the device-side domain flag back (one host sync); pass False on a timed path."""
    codes = f32c(codes)
    require_gpu(codes)
    n, K = codes.shape
    W = (K + 31) // 32
    sp = torch.empty(n, W, dtype=torch.int32, device=codes.device)
    nz = torch.empty(n, W, dtype=torch.int32, device=codes.device)
    bad = torch.zeros(1, dtype=torch.int32, device=codes.device)
    check(lib().cmh_pack_codes(ptr(codes), n, K, ptr(sp), ptr(nz), ptr(bad), stream_ptr(codes.device)), "cmh_pack_codes")
    if validate and int(bad.item()):
        raise NativeError("pack_codes: hash codes must be exactly -1, 0 or +1 (sign()/argmax codes)")
    return sp, nz


def unpack_codes(sign_plane, nz_plane, bits: int):
    """(sign_plane, nz_plane) int32 [n, ceil(bits/32)] -> f32 codes [n, bits] in {-1, 0, +1} (the inverse of pack_codes)."""
    require_gpu(sign_plane, nz_plane)
    sp, nz = sign_plane.contiguous(), nz_plane.contiguous()
    n, W = sp.shape
    if sp.dtype != torch.int32 or nz.dtype != torch.int32 or tuple(nz.shape) != (n, W) or W != (int(bits) + 31) // 32:
        raise NativeError(f"unpack_codes: planes {tuple(sp.shape)} / {tuple(nz.shape)} do not hold {bits}-bit codes")
    out = torch.empty(n, int(bits), dtype=torch.float32, device=sp.device)
    if n:
        check(lib().cmh_unpack_codes(ptr(sp), ptr(nz), n, int(bits), ptr(out), stream_ptr(sp.device)), "cmh_unpack_codes")
    return out


def map_mean(ap):
    """f32 [Q] per-query APs -> their mean as the ranking kernel forms it: a sequential f32 sum in query order, / Q (0-dim tensor)."""
    ap = f32c(ap)
    require_gpu(ap)
    out = torch.empty(1, dtype=torch.float32, device=ap.device)
    check(lib().cmh_map_mean(ptr(ap), ap.numel(), ptr(out), stream_ptr(ap.device)), "cmh_map_mean")
    return out[0]


def pack_labels(labels):
    labels = f32c(labels)
    require_gpu(labels)
    n, Cn = labels.shape
    LW = (Cn + 31) // 32
    out = torch.empty(n, LW, dtype=torch.int32, device=labels.device)
    bad = torch.zeros(1, dtype=torch.int32, device=labels.device)
    check(lib().cmh_pack_labels(ptr(labels), n, Cn, ptr(out), ptr(bad), stream_ptr(labels.device)), "cmh_pack_labels")
    if int(bad.item()):
        raise NativeError("pack_labels: labels must be non-negative (multi-hot)")
    return out


def hamming_dist(q_planes, r_planes, bits):
    (qs, qn), (rs, rn) = q_planes, r_planes
    Q, N = qs.shape[0], rs.shape[0]
    W = (bits + 31) // 32
    fit("hamming_dist", (qs, (Q, W)), (qn, (Q, W)), (rs, (N, W)), (rn, (N, W)))
    out = torch.empty(Q, N, dtype=torch.float32, device=qs.device)
    check(lib().cmh_hamming_dist(ptr(qs), ptr(qn), ptr(rs), ptr(rn), Q, N, bits, ptr(out), stream_ptr(qs.device)),
          "cmh_hamming_dist")
    return out


def calc_neighbor(la, lb, classes):
    A, B = la.shape[0], lb.shape[0]
    fit("calc_neighbor", (la, (A, (classes + 31) // 32)), (lb, (B, (classes + 31) // 32)))
    out = torch.empty(A, B, dtype=torch.float32, device=la.device)
    check(lib().cmh_calc_neighbor(ptr(la), ptr(lb), A, B, classes, ptr(out), stream_ptr(la.device)), "cmh_calc_neighbor")
    return out


def hamming_map(q_planes, q_lab, r_planes, r_lab, bits, classes, topk=None, tie_order=TIE_REFERENCE,
                want_perm=False, depth_limit=-1):
    """-> (map 0-dim f32 tensor, ap [Q] f32, perm [Q,N] int32 or None)"""
    (qs, qn), (rs, rn) = q_planes, r_planes
    dev = qs.device
    Q, N = qs.shape[0], rs.shape[0]
    W, LW = (bits + 31) // 32, (classes + 31) // 32
    fit("hamming_map", (qs, (Q, W)), (qn, (Q, W)), (rs, (N, W)), (rn, (N, W)), (q_lab, (Q, LW)), (r_lab, (N, LW)))
    ap = torch.empty(Q, dtype=torch.float32, device=dev)
    mp = torch.empty(1, dtype=torch.float32, device=dev)
    perm = torch.empty(Q, N, dtype=torch.int32, device=dev) if want_perm else None
    need = lib().cmh_map_workspace_bytes(Q, N, bits, tie_order)
    ws = workspace(need, dev, "map")
    check(lib().cmh_hamming_map(ptr(qs), ptr(qn), ptr(q_lab), ptr(rs), ptr(rn), ptr(r_lab), Q, N, bits, classes,
                                0 if topk is None else int(topk), tie_order, depth_limit, ptr(ap), ptr(mp), ptr(perm),
                                ptr(ws), ws.numel(), stream_ptr(dev)), "cmh_hamming_map")
    return mp[0], ap, perm


# ------------------------------------------------------------------------------------------ losses
def fit(what, *pairs):
    """pairs of (tensor, expected shape): the kernels take their sizes from ONE operand, so every other operand is checked here
    (a mismatched operand would be read out of bounds)."""
    for t, shape in pairs:
        if t is not None and tuple(t.shape) != tuple(shape):
            raise NativeError(f"{what}: operand of shape {tuple(t.shape)} where {tuple(shape)} is expected")


def dsph_hyp_loss(x, y, label, proxies, threshold, alpha):
    x, y, label, proxies = f32c(x), f32c(y), f32c(label), f32c(proxies)
    require_gpu(x, y, label, proxies)
    B, K = x.shape
    Cn = label.shape[1]
    fit("dsph_hyp_loss", (y, (B, K)), (label, (B, Cn)), (proxies, (Cn, K)))
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = workspace(lib().cmh_loss_workspace_bytes(B, K, Cn), x.device, "loss")
    check(lib().cmh_dsph_hyp_loss(ptr(x), ptr(y), ptr(label), ptr(proxies), B, K, Cn, float(threshold), float(alpha),
                                  ptr(out), ptr(ws), ws.numel(), stream_ptr(x.device)), "cmh_dsph_hyp_loss")
    return out[0]


def msl_loss(feats, labels, feat2=None):
    """DMsH-LN multi-similarity loss (train/DMsH_LN/MSLOSS.py:13-55) -> loss 0-dim"""
    feats, labels = f32c(feats), f32c(labels)
    feat2 = None if feat2 is None else f32c(feat2)
    require_gpu(feats, labels, feat2)
    B, K = feats.shape
    fit("msl_loss", (labels, (B, labels.shape[1])))
    if feat2 is not None:
        fit("msl_loss", (feat2, (B, K)))
    out = torch.empty(1, dtype=torch.float32, device=feats.device)
    ws = workspace(lib().cmh_msl_workspace_bytes(B), feats.device, "loss")
    check(lib().cmh_msl_loss(ptr(feats), ptr(feat2), ptr(labels), B, K, labels.shape[1], ptr(out), ptr(ws), ws.numel(),
                             stream_ptr(feats.device)), "cmh_msl_loss")
    return out[0]


def msl_loss_backward(feats, labels, feat2, dloss):
    """-> (dfeats, dfeat2 | None); with feat2 = None the gradients of both roles of feats are summed into dfeats"""
    B, K = feats.shape
    dfeats = torch.empty_like(feats)
    dfeat2 = None if feat2 is None else torch.empty_like(feat2)
    ws = workspace(lib().cmh_msl_workspace_bytes(B), feats.device, "loss")
    check(lib().cmh_msl_loss_backward(ptr(feats), ptr(feat2), ptr(labels), B, K, labels.shape[1], ptr(f32c(dloss).reshape(1)),
                                      ptr(dfeats), ptr(dfeat2), ptr(ws), ws.numel(), stream_ptr(feats.device)), "cmh_msl_loss_backward")
    return dfeats, dfeat2


def spl_loss(a, b, labels, temperature, delta):
    """DHaPH self-paced contrastive loss (train/DHaPH/MSLoss.py:13-33); b = None: a against itself -> loss 0-dim"""
    a, labels = f32c(a), f32c(labels)
    b = None if b is None else f32c(b)
    require_gpu(a, labels, b)
    B, K = a.shape
    fit("spl_loss", (labels, (B, labels.shape[1])), (b, (B, K)))
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    ws = workspace(lib().cmh_spl_workspace_bytes(B), a.device, "loss")
    check(lib().cmh_spl_loss(ptr(a), ptr(b), ptr(labels), B, K, labels.shape[1], float(temperature), float(delta), ptr(out), ptr(ws),
                             ws.numel(), stream_ptr(a.device)), "cmh_spl_loss")
    return out[0]


def spl_loss_backward(a, b, labels, temperature, delta, dloss):
    """-> (da, db | None); with b = None the gradients of both roles of a are summed into da"""
    B, K = a.shape
    da = torch.empty_like(a)
    db = None if b is None else torch.empty_like(b)
    ws = workspace(lib().cmh_spl_workspace_bytes(B), a.device, "loss")
    check(lib().cmh_spl_loss_backward(ptr(a), ptr(b), ptr(labels), B, K, labels.shape[1], float(temperature), float(delta),
                                      ptr(f32c(dloss).reshape(1)), ptr(da), ptr(db), ptr(ws), ws.numel(), stream_ptr(a.device)),
          "cmh_spl_loss_backward")
    return da, db


def qmi_loss(img, txt, label, eps=1e-8):
    """DNpH qmi_loss (train/DNpH_TMM/loss.py:5-72, defaults) -> (loss 0-dim, sum_d [1] for the backward, packed labels)"""
    img, txt, label = f32c(img), f32c(txt), f32c(label)
    require_gpu(img, txt, label)
    B, K = img.shape
    Cn = label.shape[1]
    fit("qmi_loss", (txt, (B, K)), (label, (B, Cn)))
    packed = pack_labels(label)
    out = torch.empty(1, dtype=torch.float32, device=img.device)
    sum_d = torch.empty(1, dtype=torch.float32, device=img.device)
    ws = workspace(lib().cmh_qmi_workspace_bytes(B), img.device, "loss")
    check(lib().cmh_qmi_loss(ptr(img), ptr(txt), ptr(packed), B, K, Cn, float(eps), ptr(out), ptr(sum_d), ptr(ws), ws.numel(),
                             stream_ptr(img.device)), "cmh_qmi_loss")
    return out[0], sum_d, packed


def qmi_loss_backward(img, txt, packed, classes, sum_d, dloss, eps=1e-8):
    B, K = img.shape
    dimg, dtxt = torch.empty_like(img), torch.empty_like(txt)
    ws = workspace(lib().cmh_qmi_workspace_bytes(B), img.device, "loss")
    check(lib().cmh_qmi_loss_backward(ptr(img), ptr(txt), ptr(packed), B, K, int(classes), float(eps), ptr(sum_d), ptr(f32c(dloss).reshape(1)),
                                      ptr(dimg), ptr(dtxt), ptr(ws), ws.numel(), stream_ptr(img.device)), "cmh_qmi_loss_backward")
    return dimg, dtxt


def dchmt_loss(img, txt, label, output_dim, similarity="euclidean", loss_type="l2", vartheta=0.5, sim_threshold=0.1):
    img, txt, label = f32c(img), f32c(txt), f32c(label)
    require_gpu(img, txt, label)
    B, D = img.shape
    Cn = label.shape[1]
    fit("dchmt_loss", (txt, (B, D)), (label, (B, Cn)))
    out = torch.empty(1, dtype=torch.float32, device=img.device)
    ws = workspace(lib().cmh_loss_workspace_bytes(B, D, Cn), img.device, "loss")
    sim = {"euclidean": 0, "cosine": 1}[similarity]
    lt = {"l1": 1, "l2": 2}[loss_type]
    check(lib().cmh_dchmt_loss(ptr(img), ptr(txt), ptr(label), B, D, Cn, int(output_dim), sim, lt, float(vartheta),
                               float(sim_threshold), ptr(out), ptr(ws), ws.numel(), stream_ptr(img.device)),
          "cmh_dchmt_loss")
    return out[0]


# ------------------------------------------------------------------------------------------ DNPH / TwDH
def batchnorm1d_train(x, w, b, eps=1e-5):
    x, w, b = f32c(x), f32c(w), f32c(b)
    require_gpu(x, w, b)
    B, d = x.shape
    fit("batchnorm1d_train", (w, (d,)), (b, (d,)))
    y = torch.empty_like(x)
    check(lib().cmh_batchnorm1d_train(ptr(x), ptr(w), ptr(b), float(eps), ptr(y), B, d, stream_ptr(x.device)),
          "cmh_batchnorm1d_train")
    return y


def twdh_targets(label, center, random_center):
    label, center, random_center = f32c(label), f32c(center), f32c(random_center)
    require_gpu(label, center, random_center)
    B, Cn = label.shape
    K = center.shape[1]
    fit("twdh_targets", (center, (Cn, K)), (random_center, (K,)))
    code = torch.empty(B, K, dtype=torch.float32, device=label.device)
    check(lib().cmh_twdh_targets(ptr(label), ptr(center), ptr(random_center), ptr(code), B, Cn, K,
                                 stream_ptr(label.device)), "cmh_twdh_targets")
    return code


def twdh_loss(p_img, p_txt, target):
    """-> (nce, quan) 0-dim tensors for one code length."""
    p_img, p_txt, target = f32c(p_img), f32c(p_txt), f32c(target)
    require_gpu(p_img, p_txt, target)
    B, K = target.shape
    fit("twdh_loss", (p_img, (B, 2 * K)), (p_txt, (B, 2 * K)))
    out = torch.empty(2, dtype=torch.float32, device=p_img.device)
    ws = workspace(256, p_img.device, "loss")
    check(lib().cmh_twdh_loss(ptr(p_img), ptr(p_txt), ptr(target), B, K, ptr(out), ptr(ws), ws.numel(),
                              stream_ptr(p_img.device)), "cmh_twdh_loss")
    return out[0], out[1]


def dnph_loss(hash_img, hash_txt, pre_img, pre_txt, label, proxies, noise_img=None, noise_txt=None, margin=1.0,
              noise_weight=0.1):
    """-> (loss, p_loss + d_loss, noise) 0-dim tensors."""
    ts = [f32c(t) for t in (hash_img, hash_txt, pre_img, pre_txt, label, proxies)]
    ni = None if noise_img is None else f32c(noise_img)
    nt = None if noise_txt is None else f32c(noise_txt)
    require_gpu(*ts, ni, nt)
    B, K = ts[0].shape
    Cn = ts[4].shape[1]
    fit("dnph_loss", (ts[1], (B, K)), (ts[2], (B, Cn)), (ts[3], (B, Cn)), (ts[4], (B, Cn)), (ts[5], (Cn, K)), (ni, (B, K)), (nt, (B, K)))
    out = torch.empty(3, dtype=torch.float32, device=ts[0].device)
    ws = workspace(256, ts[0].device, "loss")
    check(lib().cmh_dnph_loss(*[ptr(t) for t in ts], ptr(ni), ptr(nt), B, K, Cn, float(margin), float(noise_weight),
                              ptr(out), ptr(ws), ws.numel(), stream_ptr(ts[0].device)), "cmh_dnph_loss")
    return out[0], out[1], out[2]


# ------------------------------------------------------------------------------------------ optimiser
def bert_adam_step(entries, b1, b2, eps):
    """One fused BertAdam step over `entries` = [(param, grad, next_m, next_v, lr_scheduled, weight_decay, max_grad_norm)]
    (all f32, contiguous, on one GPU).  Updates param / next_m / next_v in place and, like clip_grad_norm_, the clipped grad."""
    if not entries:
        return
    dev = entries[0][0].device
    arr = (AdamTensor * len(entries))()
    total = 0
    for i, (p, g, m, v, lr, wd, mx) in enumerate(entries):
        require_gpu(p, g, m, v)
        for t in (p, g, m, v):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
                raise NativeError("bert_adam_step: tensors must be contiguous f32 of one size")
        # the encoder's bf16 GEMM copy of this parameter (model/base/model.py::_gemm_w), if one exists: the step rewrites it
        copy = getattr(p, "_cmh_bf16", None)
        if copy is not None and (copy.numel() != p.numel() or copy.device != p.device or copy.dtype != torch.bfloat16):
            copy = None
        arr[i] = AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(wd), float(mx),
                            None if copy is None else copy.data_ptr())
        total += p.numel()
    need = lib().cmh_bert_adam_workspace_bytes(len(entries), total)
    ws = workspace(need, dev, "adam")
    check(lib().cmh_bert_adam_step(arr, len(entries), float(b1), float(b2), float(eps), ptr(ws), ws.numel(), stream_ptr(dev)),
          "cmh_bert_adam_step")
