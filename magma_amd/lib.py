"""ctypes binding of libmagma_hip.so (the C ABI declared in include/magma_hip.h).

There is deliberately NO fallback: if the shared library is missing, or an op
is asked to run on a non-GPU tensor, we raise.  The CPU restatement in
``oracle/`` is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libmagma_hip.so"

MG_ACT_NONE, MG_ACT_RELU, MG_ACT_GELU_NEW, MG_ACT_QUICK_GELU = 0, 1, 2, 3
MG_W_ROWMAJOR, MG_W_FRAGTILED = 0, 1
MG_A_DENSE, MG_A_CONV3X3 = 0, 1
MG_AUX_NONE, MG_AUX_RELU_GATE, MG_AUX_GELU_GRAD, MG_AUX_MUL, MG_AUX_QUICK_GELU_GRAD = 0, 1, 2, 3, 4


ABI_VERSION = 6      # include/magma_hip.h MG_ABI_VERSION


class MagmaHipError(RuntimeError):
    pass


class Epilogue(C.Structure):
    _fields_ = [
        ("scale", C.c_void_p), ("bias", C.c_void_p),
        ("act", C.c_int32), ("act_after", C.c_int32),
        ("res0", C.c_void_p), ("res1", C.c_void_p), ("res2", C.c_void_p),
        ("ldr", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("out_f32", C.c_int32), ("aux_mode", C.c_int32),
        ("aux", C.c_void_p), ("ldaux", C.c_int64),
        ("aux_after", C.c_int32), ("act_n0", C.c_int32),
        ("C2", C.c_void_p), ("ldc2", C.c_int64),
        ("C8", C.c_void_p), ("ldc8", C.c_int64), ("c8_scales", C.c_void_p), ("c8_rgroups", C.c_int32), ("accumulate", C.c_int32),
        ("row_scale", C.c_void_p),
    ]


class RelayoutJob(C.Structure):          # mg_relayout_job
    _fields_ = [("w", C.c_void_p), ("scale", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_int64),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("k", C.c_int32), ("mode", C.c_int32), ("first_block", C.c_int64)]


class BnFoldJob(C.Structure):            # mg_bn_fold_job
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("var", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("eps", C.c_float), ("C", C.c_int32), ("first_block", C.c_int64)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("a_mode", C.c_int32), ("w_layout", C.c_int32),
        ("H", C.c_int32), ("Wd", C.c_int32), ("Cin", C.c_int32),
        ("zero_page", C.c_void_p),
        ("ep", Epilogue),
        ("tile_hint", C.c_int32), ("split_k", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
    ]


class SkinnyDesc(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("ldx", C.c_int64),
        ("W", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("Kp", C.c_int32), ("nt_hint", C.c_int32),
        ("ep", Epilogue),
        ("ln_colsum", C.c_void_p), ("ln_inv_d", C.c_float), ("ln_eps", C.c_float),
        ("split_n", C.c_int32), ("_pad", C.c_int32),
        ("ep_b", Epilogue),
        ("w_scale", C.c_void_p),
    ]


# every symbol include/magma_hip.h declares: (name, restype, argtypes)
_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
SYMBOLS = {
    "mg_version": (C.c_char_p, []),
    "mg_last_error": (C.c_char_p, []),
    "mg_abi_version": (C.c_int32, []),
    "mg_gemm_bf16": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "mg_gemm_workspace_bytes": (C.c_int64, [_i32, _i32, _i32]),
    "mg_gemm_skinny_bf16": (C.c_int, [C.POINTER(SkinnyDesc), _vp]),
    "mg_gemm_skinny2_bf16": (C.c_int, [C.POINTER(SkinnyDesc), C.POINTER(SkinnyDesc), _vp]),
    "mg_decode_attn_gemv_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _vp,
                                          C.POINTER(SkinnyDesc), _vp]),
    "mg_comm_unique_id": (C.c_int, [_vp]),
    "mg_comm_init": (C.c_int, [C.POINTER(C.c_void_p), _vp, _i32, _i32]),
    "mg_comm_allreduce_sum": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "mg_comm_broadcast": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "mg_comm_destroy": (C.c_int, [_vp]),
    "mg_attn_fp8_scale_stride": (_i32, [_i32]),
    "mg_rotary_split_fp8": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "mg_attn_prefill_fp8": (C.c_int, [_vp] * 7 + [_i64, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp]),
    "mg_stream_create_cu_mask": (C.c_int, [C.POINTER(C.c_void_p), _i32]),
    "mg_stream_destroy": (C.c_int, [_vp]),
    "mg_layernorm_bf16": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "mg_embedding_bf16": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i64, _i32, _vp]),
    "mg_rotary_split_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp]),
    "mg_rotary_split_train_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "mg_attn_prefill_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mg_attn_decode_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "mg_attn_decode_fused_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "mg_argmax_f32": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "mg_advance_pos": (C.c_int, [_vp, _i32, _vp]),
    "mg_sample_f32": (C.c_int, [_vp, _i64, _i32, _i32, _f32, _i32, C.c_double, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mg_sample_finish": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _i32, _vp, _i64, _i32, _vp, _i32, _i32, _vp]),
    "mg_patchify_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mg_vit_embed_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mg_attn_small_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "mg_attn_small_bwd_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mg_avgpool2_nhwc_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mg_stem_im2col_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "mg_weight_standardize_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i32, _f32, _f32, _vp]),
    "mg_im2col_nchw_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mg_maxpool3x3s2_nhwc_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mg_subsample2_nhwc_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mg_relu_mean_rows_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "mg_weight_standardize_bwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _f32, _f32, _f32, _vp]),
    "mg_maxpool3x3s2_bwd_nhwc_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mg_subsample2_bwd_nhwc_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mg_relu_mean_rows_bwd_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mg_build_labels_i64": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _vp]),
    "mg_ce_rows_f32": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, _vp]),
    "mg_ce_reduce_f32": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    # training path
    "mg_transpose_bf16": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "mg_transpose_colsum_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp]),
    "mg_head_transpose_bf16": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mg_colsum_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _vp]),
    "mg_layernorm_bwd_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _f32, _vp]),
    "mg_ce_bwd_bf16": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "mg_rotary_merge_bwd_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mg_attn_bwd_bf16": (C.c_int, [_vp] * 13 + [_i32, _i32, _i32, _i32, _i64, _vp]),
    "mg_attn_bwd_merged_bf16": (C.c_int, [_vp] * 11 + [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _vp]),
    "mg_rotary_qk_inplace_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mg_attn_fwd_rows_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _vp]),
    "mg_attn_bwd_rows_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                        _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "mg_avgpool2_bwd_nhwc_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mg_mul_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "mg_gelu_erf_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "mg_scale_rows_acc_f32": (C.c_int, [_vp, _vp, _i64, _vp, _i32, _i32, _vp]),
    "mg_add_gate_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "mg_bn_param_grad_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "mg_transpose_bn_param_grad_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mg_im2col_t_bf16": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mg_sumsq_f32": (C.c_int, [_vp, _i64, _vp, _vp]),
    "mg_adamw_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp, _f32, _vp]),
    "mg_bn_batch_fold_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "mg_bn_apply_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp]),
    "mg_bn_bwd_dz_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "mg_cast_f32_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "mg_sumsq_bf16": (C.c_int, [_vp, _i64, _vp, _vp]),
    "mg_adamw_gbf16_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp, _f32, _vp]),
    "mg_gemm_fp8": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp]),
    "mg_quantize_rows_fp8": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp]),
    "mg_mx_scale_bytes": (C.c_int64, [_i32, _i32]),
    "mg_quantize_mx_fp8": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp]),
    "mg_gemm_mx_fp8": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp]),
    "mg_debug_mx_mfma": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mg_conv_weight_relayout_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mg_bn_fold_f32": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _i32, _vp]),
    "mg_conv_weight_relayout_batch": (C.c_int, [_vp, _i32, _i64, _vp]),
    "mg_bn_fold_batch": (C.c_int, [_vp, _i32, _i64, _vp]),
    "mg_resample_u8": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "mg_crop_normalize_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load libmagma_hip.so (built in-tree by ``__graft_entry__.build()`` /
    ``make -C magma_amd/csrc``).  Raises MagmaHipError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("MAGMA_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise MagmaHipError(
            f"{path} not found: the MI355X HIP library is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C magma_amd/csrc`.")
    lib = C.CDLL(str(path))
    try:
        lib.mg_abi_version.restype = C.c_int32
        abi = int(lib.mg_abi_version())
    except AttributeError:
        abi = 1
    if abi != ABI_VERSION:
        raise MagmaHipError(f"{path} speaks C-ABI revision {abi}, this package binds revision {ABI_VERSION} "
                            "(include/magma_hip.h MG_ABI_VERSION): rebuild with `make -C magma_amd/csrc`")
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mg_last_error().decode("utf-8", "replace")
        raise MagmaHipError(f"{what or 'libmagma_hip'} failed (rc={rc}): {msg}")
