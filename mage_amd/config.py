"""Every switch of the package in one place.

Two kinds of switch exist, and both are read from the environment exactly ONCE:

* host-side switches (this module): which fusions / data layouts the Python mirror of the reference's classes asks the library for.  They
  are fields of the frozen ``Config`` object built at import from the ``MAGE_*`` variables named in ``ENV`` below; the package reads
  ``config.get()`` -- never ``os.environ`` -- at its decision points.  Tests and tuning scripts change them with ``config.override(...)``
  (a context manager that swaps the object; captured HIP graphs key on it, see ``MAGE._graph_fingerprint``).
* library-side switches (kernel selection inside libmage_hip.so): one table inside the library, filled once from the environment by the
  library itself, changed at run time with ``config.lib_option(name, value)`` (-> ``mage_set_option``; include/mage_hip.h).

INTEGRATION.md carries the table of all of them.  Defaults are the fast paths; every switch turns a fusion or a kernel OFF (A/B tests,
bisecting), none is needed for correct results.
"""
from __future__ import annotations

import contextlib
import dataclasses
import os
from typing import Iterator

__all__ = ["Config", "ENV", "get", "override", "lib_option", "lib_options", "lib_flag", "LIB_OPTIONS"]


@dataclasses.dataclass(frozen=True)
class Config:
    # ---- decoder stack (modules/mage_model.py)
    ln_fold: bool = True              # LayerNorm folded around the 16-bit GEMMs (producer partial sums + consumer epilogue)
    stream_16bit: bool = True         # bf16 / f16 modes keep the residual stream in that type between the blocks (False: fp32 stream + copy)
    attn_split: bool = True           # f16x3 mode: axial attention on split q / k / v rows (False: fp32 thread-per-query kernels)
    frame_table: bool = True          # conv3x3(embedding) + in_linear as a table sum (False: the convolution GEMM + in_linear)
    auto_graph: bool = True           # one-clip calls replay a captured HIP graph
    # ---- VQ-VAE (modules/vqvae_model.py)
    encode_split: bool = True         # encoder convolutions on f16x3 split operands outside precision 'fp32'
    decode_table: bool = True         # f4 decode: first 3x3 convolution as a table sum
    decode_split: bool = True         # f16x3 / bf16x3 precision: f4 decode stack on split operands (False: the exact-fp32 decode)
    decode_taps8: bool = True         # f4 / f8 decode 3x3 convolutions in the padded-taps form
    decode_resblock_fusion: bool = True
    decode_head_fusion: bool = True
    decode_phase_merge: bool = True
    decode_relu_fold: bool = True     # f8 decode: a block's leading ReLU taken on the operand fragments of its first 1x1 convolution
    # ---- training path (modules/mage_train*.py)
    train_f32_branch: bool = False    # fp32 branch rows / LayerNorm-output gradients in bf16 training (the round-2 form)
    train_wgrad_transpose: bool = False   # weight gradients through transposed copies + split-K (False: mage_gemm_tn)
    train_emit: bool = True           # backward kernels emit the next GEMM's bf16 operand (False: separate casts)
    train_dual: bool = True           # c_fc writes pre-activation + activated rows from one tile
    train_taps: bool = True           # frame convolution forward / backward in the padded-taps form
    train_enc_split: bool = True      # text / motion-anchor encoders on f16x3 split operands in bf16 training (False: exact fp32)


# field -> (environment variable, value of the field when the variable is set to a non-empty, non-"0" string)
ENV = {
    "ln_fold": ("MAGE_NO_LN_FOLD", False),
    "stream_16bit": ("MAGE_STREAM_FP32", False),
    "attn_split": ("MAGE_ATTN_SPLIT_FP32", False),
    "frame_table": ("MAGE_NO_FRAME_TABLE", False),
    "auto_graph": ("MAGE_NO_AUTO_GRAPH", False),
    "encode_split": ("MAGE_ENCODE_FP32", False),
    "decode_table": ("MAGE_NO_DECODE_TABLE", False),
    "decode_split": ("MAGE_DECODE_FP32", False),
    "decode_taps8": ("MAGE_DECODE_NO_TAPS8", False),
    "decode_resblock_fusion": ("MAGE_DECODE_NO_RESBLOCK_FUSION", False),
    "decode_head_fusion": ("MAGE_DECODE_NO_HEAD_FUSION", False),
    "decode_phase_merge": ("MAGE_DECODE_NO_PHASE_MERGE", False),
    "decode_relu_fold": ("MAGE_DECODE_NO_RELU_FOLD", False),
    "train_f32_branch": ("MAGE_TRAIN_F32_BRANCH", True),
    "train_wgrad_transpose": ("MAGE_WGRAD_TRANSPOSE", True),
    "train_emit": ("MAGE_TRAIN_NO_EMIT", False),
    "train_dual": ("MAGE_TRAIN_NO_DUAL", False),
    "train_taps": ("MAGE_TRAIN_NO_TAPS", False),
    "train_enc_split": ("MAGE_TRAIN_ENC_FP32", False),
}

# the library-side options (mage_set_option; struct MageOptions in csrc/common.h); each is read by the library from MAGE_<NAME> once
LIB_OPTIONS = ("gemm_no_4w", "gemm_no_4h", "gemm_4h_plain", "gemm4_train_forms", "gemm_no_8phase", "gemm_no_taps8", "gemm_no_narrow", "gemm_no_narrow_few", "gemm_no_small",
               "gemm_small_m", "gemm_stagger_groups", "gemm_stagger_percent", "gemm_stagger_forced",
               "gemm4_stagger_groups", "gemm4_stagger_percent", "attn_no_mfma", "attn_no_fewq", "vq_no_mfma", "conv_no_tile")


def _set(v) -> bool:
    return bool(v) and v != "0"


def _from_env() -> Config:
    kw = {}
    for field, (var, when_set) in ENV.items():
        if _set(os.environ.get(var)):
            kw[field] = when_set
    return Config(**kw)


_CONFIG = _from_env()          # the one read of the environment


def get() -> Config:
    return _CONFIG


@contextlib.contextmanager
def override(**kw) -> Iterator[Config]:
    """Swap the host-side configuration inside a ``with`` block (tests, tuning scripts)."""
    global _CONFIG
    saved = _CONFIG
    _CONFIG = dataclasses.replace(saved, **kw)
    try:
        yield _CONFIG
    finally:
        _CONFIG = saved


# Python-side mirror of the library's option table: read through mage_get_option ONCE, refreshed by every change made through this module
# (lib_option / set_lib_option) -- MAGE._graph_fingerprint and VectorQuantizedVAE._bottleneck ask per call / per block, on the host-bound path
# (ADVICE r5).  Code that calls mage_set_option directly (C hosts) must call invalidate_lib_options() if it shares the process with this package.
_LIB_CACHE = None


def invalidate_lib_options() -> None:
    global _LIB_CACHE
    _LIB_CACHE = None


def _lib_cache() -> dict:
    global _LIB_CACHE
    if _LIB_CACHE is None:
        import ctypes as C
        from . import _lib
        l = _lib.load()
        out = {}
        for name in LIB_OPTIONS:
            v = C.c_int32(0)
            _lib.check(l.mage_get_option(name.encode(), C.byref(v)), l)
            out[name] = int(v.value)
        _LIB_CACHE = out
    return _LIB_CACHE


def lib_options() -> dict:
    """Current values of the library-side options (mage_get_option)."""
    return dict(_lib_cache())


def lib_flag(name: str) -> int:
    """One library-side option (mage_get_option)."""
    c = _lib_cache()
    if name not in c:
        raise KeyError(f"unknown library option '{name}' (mage_amd.config.LIB_OPTIONS)")
    return c[name]


def set_lib_option(name: str, value: int) -> None:
    """mage_set_option + refresh of the mirror (the library validates the name and the value)."""
    from . import _lib
    l = _lib.load()
    try:
        _lib.check(l.mage_set_option(name.encode(), int(value)), l)
    finally:
        invalidate_lib_options()


@contextlib.contextmanager
def lib_option(name: str, value: int) -> Iterator[None]:
    """Set one library-side option inside a ``with`` block (mage_set_option), restoring the previous value afterwards."""
    old = lib_flag(name)
    set_lib_option(name, value)
    try:
        yield
    finally:
        set_lib_option(name, old)
