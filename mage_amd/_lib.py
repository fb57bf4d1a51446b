"""ctypes binding of libmage_hip.so (the C ABI in include/mage_hip.h).

The product path has no fallback: if the shared library is missing or the device
is not a gfx950 GPU, ``lib()`` raises and every op fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAGE_HIP_LIB", os.path.join(_HERE, "lib", "libmage_hip.so"))   # override: kernel-tuning builds only

F32, BF16, BF16X3, F16X3, F16 = 0, 1, 2, 3, 4  # BF16X3 / F16X3: split-precision operands; F16: single-pass half operands (include/mage_hip.h)
ACT_NONE, ACT_RELU, ACT_QUICKGELU, ACT_GELU_ERF, ACT_TANH, ACT_QUICKGELU_GRAD = 0, 1, 2, 3, 4, 5
ABI_VERSION = 8

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("M", i32), ("N", i32), ("K", i32),
        ("A", vp), ("W", vp), ("Y", vp),
        ("lda", i32), ("ldy", i32), ("y_dtype", i32),
        ("out_h", i32), ("out_w", i32), ("in_h", i32), ("in_w", i32),
        ("a_img_stride", i32), ("a_off", i32),
        ("taps_h", i32), ("taps_w", i32), ("cin", i32), ("stride", i32),
        ("dy0", i32), ("dx0", i32), ("dys", i32), ("dxs", i32),
        ("y_img_stride", i32), ("y_mul_y", i32), ("y_mul_x", i32), ("y_off", i32),
        ("bias", vp), ("scale", vp), ("shift", vp),
        ("act", i32),
        ("rowadd", vp), ("rowadd_div", i32), ("rowadd_mod", i32),
        ("residual", vp), ("ldr", i32), ("res_dtype", i32),
        ("post_relu", i32), ("ldw", i32), ("n_split", i32),
        ("a_split_stride", i64), ("w_split_stride", i64), ("y_split_stride", i64),
        ("y2", vp), ("ldy2", i32), ("res_half", i32), ("ln_part", vp), ("ln_stats", vp), ("ln_colsum", vp), ("a_half", i32), ("ln_eps", C.c_float),
        ("head_w", vp), ("head_phases", i32), ("a_relu", i32), ("ln_part_rows", i64),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("q", vp), ("k", vp), ("v", vp), ("out", vp),
        ("ldq", i32), ("ldk", i32), ("ldv", i32), ("ldo", i32),
        ("n_seq", i32), ("inner", i32), ("nq", i32), ("nk", i32), ("n_head", i32),
        ("q_outer_stride", i32), ("q_axis_stride", i32), ("kv_outer_stride", i32), ("kv_axis_stride", i32),
        ("causal", i32), ("kv_len", vp), ("kv_len_div", i32), ("scale", f32), ("out_split", i32), ("drop_p", f32), ("drop_seed", C.c_uint64),
        ("o_outer_stride", i32), ("o_axis_stride", i32),
    ]


# name -> (restype, argtypes); every symbol include/mage_hip.h declares
SIGNATURES = {
    "mage_abi_version": (C.c_int, []),
    "mage_last_error": (C.c_char_p, []),
    "mage_init": (C.c_int, [C.c_int]),
    "mage_check_device_errors": (C.c_int, [vp]),
    "mage_set_option": (C.c_int, [C.c_char_p, i32]),
    "mage_get_option": (C.c_int, [C.c_char_p, C.POINTER(i32)]),
    "mage_gemm": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "mage_gemm_is_small": (C.c_int, [i32, i32, i32]),
    "mage_ln_stats": (C.c_int, [vp, i64, i32, i32, f32, vp, vp]),
    "mage_row_stats": (C.c_int, [vp, i32, i64, i32, i64, f32, vp, vp]),
    "mage_groupnorm_bwd": (C.c_int, [vp, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, i64, i64, vp, vp, vp, vp, vp, vp]),
    "mage_adain_bwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "mage_reparam_kl_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "mage_mse_bwd": (C.c_int, [vp, i64, vp, i64, i64, i32, vp, vp, i64, vp]),
    "mage_transpose_colsum": (C.c_int, [vp, i64, vp, i64, i64, i64, i32, i32, i64, i64, vp, i32, vp]),
    "mage_maxpool2_bwd": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "mage_upsample2_bwd": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "mage_layernorm": (C.c_int, [vp, vp, vp, vp, i32, i64, i32, f32, vp]),
    "mage_split": (C.c_int, [vp, i64, vp, i64, i64, i32, i32, vp]),
    "mage_attention": (C.c_int, [C.POINTER(AttnDesc), vp]),
    "mage_embedding": (C.c_int, [vp, vp, vp, i32, i64, i32, i32, i32, i64, i64, i64, i64, i64, vp]),
    "mage_table_conv": (C.c_int, [vp, i64, i32, i32, i32, i32, vp, i32, i32, i32, vp, vp, i32, vp, i64, i32, vp, i32, i64, i64, i64, i64, vp]),
    "mage_resblock_table": (C.c_int, [vp, i64, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, i64, i64, i64, i64, vp]),
    "mage_resblock_rows": (C.c_int, [vp, i64, vp, vp, vp, vp, vp, i64, i32, vp, i64, i64, i32, i32, i32, i64, i64, i64, vp]),
    "mage_vq_nearest": (C.c_int, [vp, vp, vp, i64, i32, i32, vp, vp, vp]),
    "mage_vq_prepare": (C.c_int, [vp, i32, i32, vp, vp, vp]),
    "mage_argmax": (C.c_int, [vp, i64, i32, i64, i64, i64, i64, vp, i64, i64, vp, vp]),
    "mage_cross_entropy": (C.c_int, [vp, vp, i64, i32, vp, vp, vp]),
    "mage_conv_in": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "mage_split_rows": (C.c_int, [vp, i64, vp, i64, i32, i32, i32, i64, i64, i64, i64, i64, vp, vp]),
    "mage_conv_out": (C.c_int, [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mage_convt_fold_tanh": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "mage_maxpool2": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mage_upsample2": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "mage_relu": (C.c_int, [vp, vp, i32, i64, vp]),
    "mage_cast": (C.c_int, [vp, i32, vp, i32, i64, vp]),
    "mage_adain": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "mage_add_scaled_rowvec": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "mage_row_affine": (C.c_int, [vp, vp, vp, i64, i32, i32, i32, vp]),
    "mage_caption_mask": (C.c_int, [vp, i32, i32, i64, vp, vp, vp]),
    "mage_copy2d": (C.c_int, [vp, i64, vp, i64, i64, i64, vp]),
    "mage_groupnorm_silu": (C.c_int, [vp, i64, i64, i32, i32, i32, i32, vp, vp, f32, vp, vp, i32, vp]),
    "mage_groupnorm_act": (C.c_int, [vp, i64, i64, i32, i32, i32, i32, vp, vp, f32, vp, vp, i32, vp, i32, i64, i64, vp]),
    "mage_reparam_kl": (C.c_int, [vp, vp, vp, vp, vp, i32, i64, vp]),
    "mage_mse": (C.c_int, [vp, i64, vp, i64, i64, i32, vp, vp, vp]),
    # training path
    "mage_transpose": (C.c_int, [vp, i32, i64, vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, i64, i64, i32, i32, i32, vp]),
    "mage_bn_colreduce": (C.c_int, [i32, vp, vp, vp, vp, vp, i64, i32, vp, i32, vp]),
    "mage_bn_apply": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "mage_bn_bwd_apply": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "mage_convt_unfold_tanh_bwd": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "mage_gemm_tn": (C.c_int, [vp, i64, vp, i64, i64, i32, i32, i32, i64, vp, vp, vp]),
    "mage_colsum": (C.c_int, [vp, i64, i64, i32, vp, i32, vp]),
    "mage_row_sum": (C.c_int, [vp, i32, i64, i64, i32, vp, i32, vp]),
    "mage_sum_partials": (C.c_int, [vp, i64, i32, i64, vp, i32, vp]),
    "mage_layernorm_bwd": (C.c_int, [vp, vp, vp, i32, vp, vp, i32, i64, i32, f32, i32, vp, f32, C.c_uint64, vp]),
    "mage_act": (C.c_int, [vp, vp, i32, i64, i32, vp]),
    "mage_act_bwd": (C.c_int, [vp, vp, vp, i32, i64, i32, vp]),
    "mage_cross_entropy_bwd": (C.c_int, [vp, vp, i64, i32, vp, vp, i32, vp]),
    "mage_embedding_bwd": (C.c_int, [vp, vp, i32, vp, i64, i32, i32, i64, i64, i64, i64, vp, i64, vp]),
    "mage_group_rowsum": (C.c_int, [vp, i32, i64, i32, i64, i64, vp, i64, vp, i32, vp]),
    "mage_attention_bwd": (C.c_int, [C.POINTER(AttnDesc), vp, vp, vp, vp, i32, i32, i32, vp]),
    "mage_dropout": (C.c_int, [vp, i32, vp, i32, i64, f32, C.c_uint64, i32, vp]),
    "mage_dropout_add": (C.c_int, [vp, i32, vp, vp, vp, i64, f32, C.c_uint64, vp]),
    "mage_dropout_add_layernorm": (C.c_int, [vp, i32, vp, vp, vp, vp, vp, i32, i64, i32, f32, f32, C.c_uint64, vp]),
    "mage_adam": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, f32, vp]),
}

_lib: Optional[C.CDLL] = None
_inited_devices = set()


class MageHipError(RuntimeError):
    pass


def load(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the library and bind every symbol (no GPU needed: used by the CPU test suite)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise MageHipError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C mage_amd/csrc`). There is no CPU/PyTorch fallback for the MAGE product path.")
    # torch bundles its own libamdhip64.so.7; import it FIRST so that this library binds to the HIP runtime
    # that owns torch's streams and allocations (two HIP runtimes in one process do not see each other's devices).
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale / missing a symbol
        fn.restype, fn.argtypes = res, args
    if lib.mage_abi_version() != ABI_VERSION:
        raise MageHipError(f"libmage_hip.so ABI {lib.mage_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def lib(device_index: int = 0) -> C.CDLL:
    """Library handle ready for compute on ``device_index`` (runs mage_init once per device)."""
    l = load()
    if device_index not in _inited_devices:
        check(l.mage_init(int(device_index)), l)
        _inited_devices.add(device_index)
    return l


def check(code: int, l: Optional[C.CDLL] = None) -> None:
    if code == 0:
        return
    l = l or load()
    msg = (l.mage_last_error() or b"").decode(errors="replace")
    if code == -1:
        raise ValueError(f"mage_hip: {msg}")
    raise MageHipError(f"mage_hip error {code}: {msg}")
