"""MI355X-native MAGE (drop-in for the reference's modules/mage_model.py).

Same class names, constructor kwargs, ``forward`` / ``autoregressive_generate`` signatures and
state_dict layout as the reference (mage_model.py:15-693).  ``nn.Linear`` /
``nn.MultiheadAttention`` / ``nn.LayerNorm`` objects are parameter containers only; all arithmetic is
libmage_hip.so: channels-last activations ([B, L, H, W, C] is a row-major [tokens, C] matrix), axial
attention over strided row sets (no permute/contiguous copies -- the reference spends 19 % of its time
there), LayerNorm / bias / QuickGELU / residual / positional tables fused around MFMA GEMMs.

Precision: ``set_precision('fp32')`` (default; exact-fp32 MFMA, the parity mode), ``'bf16'``
(bf16 MFMA with fp32 accumulation for the decoder stack and VQ-VAE decode; the residual stream,
LayerNorm, softmax, logits, the once-per-clip prologue and the VQ-VAE encode + quantiser stay fp32), or the
FAST PARITY modes ``'f16x3'`` / ``'bf16x3'``: everything as in fp32 mode except that the decoder's Linear layers and
the frame convolution multiply split-precision operands (x ~ hi + lo in two 16-bit pieces; three f16 / bf16 MFMA
products per K slab, fp32 accumulation -- include/mage_hip.h): fp32-class logits at 3/16 of the exact-fp32 MFMA cost.
"""
from __future__ import annotations

import gc
from collections import OrderedDict
from math import exp
from typing import Dict, Optional

import torch
from torch import nn

from .. import config, ops
from ..utils.util import default, instantiate_from_config, zero_module
from .vqvae_model import VectorQuantizedVAE, _Derived, _conv_w, _no_torch_forward, weights_frozen

__all__ = ["QuickGELU", "AxialAttentionBlock", "TransformerBlock", "MAEncoder", "TransformerTextEncoder", "BasicBlock",
           "ADAIN2D", "FlatAxialDecoder", "PIDControl", "MAGE"]

F32 = torch.float32
BF16 = torch.bfloat16
F16 = torch.float16
HALF_TYPES = (BF16, F16)                      # the 16-bit compute dtypes: the same kernels on v_mfma_f32_16x16x32_bf16 / _f16


def _sfx(dt: torch.dtype) -> str:
    return ".f32" if dt == F32 else ".bf16" if dt == BF16 else ".f16"


def _wdt(d: Dict[str, torch.Tensor], name: str, dt: torch.dtype) -> torch.Tensor:
    """The copy of the fp32 weight d[name + '.f32'] in the compute dtype (the f16 copies are built on first use: most models never ask)."""
    w = d.get(name + _sfx(dt))
    if w is None:
        w = d[name + _sfx(dt)] = d[name + ".f32"].to(dt)
    return w


# name -> (compute dtype of the decoder stack, split kind).  'f16' = the bf16 mode's kernels, schedules and data flow with IEEE half operands
# (11 significand bits instead of 8, same MFMA rate): every rounding of 'bf16' mode is 8x smaller; values must stay inside +-65504.
PRECISIONS = {"fp32": (F32, 0), "bf16": (BF16, 0), "f16": (F16, 0), "bf16x3": (F32, ops.BF16X3), "f16x3": (F32, ops.F16X3)}


class FrameTokens:
    """Frames handed to the decoder as TOKEN IDS [n_img, hw] instead of features: conv3x3(embedding(ids)) + positions and the
    decoder's in_linear are one table sum straight into the residual stream (MAGE._frame_tables, mage_table_conv)."""

    def __init__(self, tokens: torch.Tensor, T2: torch.Tensor, P2: torch.Tensor, R: int):
        self.tokens, self.T2, self.P2, self.R = tokens.contiguous(), T2, P2, R

    def fill(self, x: torch.Tensor, *, n_img: int, per_clip: int, P: int, y_off: int, tp: torch.Tensor) -> None:
        """Rows (clip b, slot y_off/hw + l) of x [B*P*hw, C] (fp32, or the bf16 stream) <- in_linear(conv(emb(ids[b, l])) + H/W positions)
        + T positions."""
        hw = self.R * self.R
        ops.table_conv(self.tokens, self.T2, x, n_img=n_img, H=self.R, W=self.R, pos=self.P2, rowadd=tp, rowadd_div=hw, rowadd_mod=P,
                       ldy=x.shape[1], group=per_clip * hw, y_group_stride=P * hw, y_off=y_off)


def _wsplit(d: Dict[str, torch.Tensor], name: str, kind: int) -> torch.Tensor:
    """Split-precision copy of the fp32 weight d[name + '.f32'] ([N, K] -> [N, 2K] 16-bit pieces), built on first use."""
    key = f"{name}.s{kind}"
    w = d.get(key)
    if w is None:
        w = d[key] = ops.split(d[name + ".f32"].reshape(d[name + ".f32"].shape[0], -1), kind)
    return w


def _lin_fp32(a, d, name, y, *, M, N, K, sk=0, lo=0, hi=None, a_split=None, **kw):
    """fp32-class Linear of the once-per-clip prologue (text encoder, motion-anchor encoder): y = a @ W[lo:hi]^T + b[lo:hi] (+ epilogue).
    sk = 0: the exact-fp32 MFMA chain (precision 'fp32').  sk = F16X3: split-precision operands, three f16 MFMA products per K slab
    (mage_hip.h) -- 3x the rate, and a smaller error than the fp32 chain's own accumulation (profiles/r03_split_probe.txt); `a` is split
    here (one small pass) unless the producer already wrote split rows (a_split)."""
    hi = d[name + ".f32"].shape[0] if hi is None else hi
    b = d.get(name + ".b")
    b = None if b is None else b[lo:hi]
    if sk and K % 64 == 0 and kw.get("act", ops.ACT_NONE) in (ops.ACT_NONE, ops.ACT_QUICKGELU):
        a_s = a_split if a_split is not None else ops.split(a, sk)
        return ops.gemm(a_s, _wsplit(d, name, sk)[lo:hi], y, M=M, N=N, K=K, lda=2 * K, ldy=kw.pop("ldy", N), bias=b, split_kind=sk, **kw)
    return ops.gemm(a, d[name + ".f32"][lo:hi], y, M=M, N=N, K=K, lda=K, ldy=kw.pop("ldy", N), bias=b, **kw)


def _to_dt(x: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
    """fp32 rows in the decoder's compute dtype (mage_cast)."""
    return x if dt == x.dtype else ops.cast(x.contiguous(), torch.empty(x.shape, device=x.device, dtype=dt))


def _assemble(images: torch.Tensor, video: torch.Tensor) -> torch.Tensor:
    """The result of autoregressive_generate (mage_model.py:691): the input's first frame followed by the decoded frames."""
    if images.dtype == video.dtype and video.is_contiguous() and images[0, 0].is_contiguous():
        return ops.assemble_video(images, video)
    return torch.cat([images[:, 0:1].to(video.dtype), video], 1)


def _need_gpu(t: torch.Tensor, who: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{who} runs on libmage_hip.so kernels: move the model and the batch to a ROCm GPU "
                           "(there is no CPU / PyTorch fallback)")


def _pack_linear(d: Dict[str, torch.Tensor], name: str, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> None:
    w = weight.float().contiguous()
    d[name + ".f32"] = w
    d[name + ".bf16"] = w.to(BF16)
    if bias is not None:
        d[name + ".b"] = bias.float().contiguous()


def _linear(a, d, name, y, dt, *, M, N, K, **kw):
    return ops.gemm(a, _wdt(d, name, dt), y, M=M, N=N, K=K, lda=kw.pop("lda", K), ldy=kw.pop("ldy", N),
                    bias=d.get(name + ".b"), **kw)


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x) (mage_model.py:11-13); fused into the c_fc GEMM epilogue on the GPU path."""

    forward = _no_torch_forward


def _mlp(d_model: int) -> nn.Sequential:
    return nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                      ("c_proj", nn.Linear(d_model * 4, d_model))]))


class AxialAttentionBlock(nn.Module):
    """Parameter container with the reference's keys (mage_model.py:15-29)."""

    def __init__(self, d_model: int, n_head: int, dropout: float = 0.1, axial_dim: int = 1):
        super().__init__()
        self.d_model, self.n_head, self.axial_dim = d_model, n_head, axial_dim
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = _mlp(d_model)
        self.ln_2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)

    forward = _no_torch_forward


class TransformerBlock(nn.Module):
    """Parameter container (mage_model.py:72-85); ln_q / ln_kv exist in checkpoints but MAGE never applies them (:92)."""

    def __init__(self, d_model: int, n_head: int, dropout: float = 0.1):
        super().__init__()
        self.d_model, self.n_head = d_model, n_head
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_q = nn.LayerNorm(d_model)
        self.ln_kv = nn.LayerNorm(d_model)
        self.mlp = _mlp(d_model)
        self.ln_2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)

    forward = _no_torch_forward


def _pack_block(d: Dict[str, torch.Tensor], p: str, blk) -> None:
    _pack_linear(d, p + ".in_proj", blk.attn.in_proj_weight, blk.attn.in_proj_bias)
    _pack_linear(d, p + ".out_proj", blk.attn.out_proj.weight, blk.attn.out_proj.bias)
    _pack_linear(d, p + ".c_fc", blk.mlp.c_fc.weight, blk.mlp.c_fc.bias)
    _pack_linear(d, p + ".c_proj", blk.mlp.c_proj.weight, blk.mlp.c_proj.bias)
    for ln in ("ln_1", "ln_2", "ln_q", "ln_kv"):
        if hasattr(blk, ln):
            d[f"{p}.{ln}.w"] = getattr(blk, ln).weight.float().contiguous()
            d[f"{p}.{ln}.b"] = getattr(blk, ln).bias.float().contiguous()


class MAEncoder(nn.Module):
    """Motion-anchor cross-attention encoder (mage_model.py:104-117).  ``forward(x, kv)`` keeps the reference's
    sequence-first convention: x [Tq, B, C], kv [Tk, B, C] -> [Tq, B, C]; rows are addressed by stride, not permuted."""

    def __init__(self, layers: int, d_model: int, dropout: float = 0.1):
        super().__init__()
        self.d_model, self.layers = d_model, layers
        self.blocks = nn.ModuleList([TransformerBlock(d_model, d_model // 32, dropout) for _ in range(layers)])
        self.mage_plus = False          # True = the ln_q/ln_kv variant of mage_model.py:93 (MAGE+); MAGE(use_cids=False) sets it
        self.split_kind = 0             # ops.F16X3 outside precision 'fp32' (MAGE.set_precision): Linear layers on split operands (_lin_fp32)
        self._derived = _Derived(self)

    def _build(self):
        d: Dict[str, torch.Tensor] = {}
        for i, blk in enumerate(self.blocks):
            _pack_block(d, f"b{i}", blk)
        return d

    @torch.no_grad()
    def _run(self, q: torch.Tensor, kv: torch.Tensor, *, B: int, nq: int, nk: int, seq_first: bool) -> torch.Tensor:
        """q [B*nq, C] / kv [B*nk, C] fp32 rows (batch-first: row = b*n + i; seq-first: row = i*B + b)."""
        d = self._derived.get(self._build)
        Cc, dev, H = self.d_model, q.device, self.d_model // 32
        x = q.float().contiguous().clone()
        kv = kv.float().contiguous()
        inner, qo, qa, ko, ka = (B, 0, B, 0, B) if seq_first else (1, nq, 1, nk, 1)
        sk = self.split_kind if Cc % 64 == 0 else 0
        for i in range(self.layers):
            p = f"b{i}"
            qin, kvin = x, kv
            if self.mage_plus:
                qin = ops.layernorm(x, d[p + ".ln_q.w"], d[p + ".ln_q.b"], torch.empty_like(x), 1e-5)
                kvin = ops.layernorm(kv, d[p + ".ln_kv.w"], d[p + ".ln_kv.b"], torch.empty_like(kv), 1e-5)
            qp = _lin_fp32(qin, d, p + ".in_proj", torch.empty(B * nq, Cc, device=dev, dtype=F32), M=B * nq, N=Cc, K=Cc, sk=sk, lo=0, hi=Cc)
            kvp = _lin_fp32(kvin, d, p + ".in_proj", torch.empty(B * nk, 2 * Cc, device=dev, dtype=F32), M=B * nk, N=2 * Cc, K=Cc, sk=sk,
                            lo=Cc, hi=3 * Cc)
            if sk:
                # split rows straight from the producers: attention output, LayerNorm, the c_fc epilogue (QuickGELU)
                ao = ops.split_empty(B * nq, Cc, sk, dev)
                ops.attention(qp, kvp[:, :Cc], kvp[:, Cc:], ao, ldq=Cc, ldk=2 * Cc, ldv=2 * Cc, ldo=2 * Cc, n_seq=B, inner=inner,
                              nq=nq, nk=nk, n_head=H, q_outer_stride=qo, q_axis_stride=qa, kv_outer_stride=ko, kv_axis_stride=ka, out_split=sk)
                _lin_fp32(None, d, p + ".out_proj", x, M=B * nq, N=Cc, K=Cc, sk=sk, a_split=ao, residual=x, ldr=Cc)
                xn = ops.layernorm(x, d[p + ".ln_2.w"], d[p + ".ln_2.b"], ops.split_empty(B * nq, Cc, sk, dev), 1e-5, split_kind=sk)
                hdn = _lin_fp32(None, d, p + ".c_fc", ops.split_empty(B * nq, 4 * Cc, sk, dev), M=B * nq, N=4 * Cc, K=Cc, sk=sk, a_split=xn,
                                ldy=8 * Cc, act=ops.ACT_QUICKGELU, y_split=True)
                _lin_fp32(None, d, p + ".c_proj", x, M=B * nq, N=Cc, K=4 * Cc, sk=sk, a_split=hdn, residual=x, ldr=Cc)
                continue
            ao = torch.empty(B * nq, Cc, device=dev, dtype=F32)
            ops.attention(qp, kvp[:, :Cc], kvp[:, Cc:], ao, ldq=Cc, ldk=2 * Cc, ldv=2 * Cc, ldo=Cc, n_seq=B, inner=inner,
                          nq=nq, nk=nk, n_head=H, q_outer_stride=qo, q_axis_stride=qa, kv_outer_stride=ko, kv_axis_stride=ka)
            _linear(ao, d, p + ".out_proj", x, F32, M=B * nq, N=Cc, K=Cc, residual=x, ldr=Cc)
            xn = ops.layernorm(x, d[p + ".ln_2.w"], d[p + ".ln_2.b"], torch.empty_like(x), 1e-5)
            hdn = _linear(xn, d, p + ".c_fc", torch.empty(B * nq, 4 * Cc, device=dev, dtype=F32), F32, M=B * nq, N=4 * Cc, K=Cc,
                          act=ops.ACT_QUICKGELU)
            _linear(hdn, d, p + ".c_proj", x, F32, M=B * nq, N=Cc, K=4 * Cc, residual=x, ldr=Cc)
        return x

    def forward(self, x: torch.Tensor, kv: torch.Tensor, key_mask=None, need_weights=False):
        _need_gpu(x, "MAEncoder")
        if key_mask is not None:
            raise NotImplementedError("key_mask is never passed by MAGE (mage_model.py:596,657)")
        Tq, B, Cc = x.shape
        return self._run(x.reshape(Tq * B, Cc), kv.reshape(-1, Cc), B=B, nq=Tq, nk=kv.shape[0], seq_first=True).view(Tq, B, Cc)


class TransformerTextEncoder(nn.Module):
    """Text encoder (mage_model.py:180-250): token+position embedding -> LN(eps 1e-8) -> zero padded rows ->
    post-norm encoder layers (erf-GELU, key-padding mask) -> LN -> Linear.  forward(text int64 [B,S]) -> [B,S,out]."""

    def __init__(self, vocab_size: int, transformer_width: int, transformer_layers: int, output_dim: int,
                 context_length: int, padding_idx: int = 0, dropout: float = 0.1):
        super().__init__()
        self.vocab_size, self.padding_idx = vocab_size, padding_idx
        self.transformer_width, self.context_length = transformer_width, context_length
        self.transformer_layers, self.output_dim = transformer_layers, output_dim
        layer = nn.TransformerEncoderLayer(transformer_width, transformer_width // 32, dim_feedforward=transformer_width * 4,
                                           dropout=dropout, activation="gelu")
        self.transformer = nn.TransformerEncoder(layer, transformer_layers, enable_nested_tensor=False)
        self.token_embedding = nn.Embedding(vocab_size, transformer_width, padding_idx=padding_idx)
        self.positions = nn.Embedding(context_length, transformer_width)
        self.layer_norm = nn.LayerNorm(transformer_width, eps=1e-8, elementwise_affine=True)
        self.dropout = nn.Dropout(p=dropout)
        self.ln_text_final = nn.LayerNorm(transformer_width)
        self.text_projection = nn.Linear(transformer_width, output_dim)
        self.apply(self._init_weights)
        self._derived = _Derived(self)

    @staticmethod
    def _init_weights(module):
        """mage_model.py:211-221."""
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=0.02)
        elif isinstance(module, nn.MultiheadAttention):
            module.in_proj_weight.data.normal_(mean=0.0, std=0.02)
            module.out_proj.weight.data.normal_(mean=0.0, std=0.02)
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()

    def _build(self):
        d: Dict[str, torch.Tensor] = {"tok": self.token_embedding.weight.float().contiguous(),
                                      "pos": self.positions.weight.float().contiguous()}
        for n in ("layer_norm", "ln_text_final"):
            d[n + ".w"], d[n + ".b"] = getattr(self, n).weight.float().contiguous(), getattr(self, n).bias.float().contiguous()
        _pack_linear(d, "proj", self.text_projection.weight, self.text_projection.bias)
        for i, l in enumerate(self.transformer.layers):
            _pack_linear(d, f"l{i}.in_proj", l.self_attn.in_proj_weight, l.self_attn.in_proj_bias)
            _pack_linear(d, f"l{i}.out_proj", l.self_attn.out_proj.weight, l.self_attn.out_proj.bias)
            _pack_linear(d, f"l{i}.fc1", l.linear1.weight, l.linear1.bias)
            _pack_linear(d, f"l{i}.fc2", l.linear2.weight, l.linear2.bias)
            for n in ("norm1", "norm2"):
                d[f"l{i}.{n}.w"], d[f"l{i}.{n}.b"] = getattr(l, n).weight.float().contiguous(), getattr(l, n).bias.float().contiguous()
                d[f"l{i}.{n}.eps"] = getattr(l, n).eps
        return d

    @torch.no_grad()
    def forward(self, text: torch.Tensor) -> torch.Tensor:
        _need_gpu(text, "TransformerTextEncoder")
        d = self._derived.get(self._build)
        B, S = text.shape
        if S > 64 or S > self.context_length:
            raise ValueError(f"caption length {S} exceeds context_length {self.context_length} (kernel limit 64)")
        Wd, dev, H = self.transformer_width, text.device, self.transformer_width // 32
        ids = text.to(torch.int64).contiguous()
        kv_len, keep = ops.caption_mask(ids, self.padding_idx)          # key-padding lengths, row scale of the padded rows (:233-239)
        x = ops.embedding(ids, d["tok"], torch.empty(B * S, Wd, device=dev, dtype=F32))
        ops.row_affine(x, None, d["pos"], div=1, mod=S)                  # + positions[0..S)       (:227-228)
        ops.layernorm(x, d["layer_norm.w"], d["layer_norm.b"], x, self.layer_norm.eps)
        ops.row_affine(x, keep, None)                                    # zero padded rows          (:233-235)
        sk = getattr(self, "split_kind", 0) if Wd % 64 == 0 else 0
        for i in range(self.transformer_layers):
            p = f"l{i}"
            qkv = _lin_fp32(x, d, p + ".in_proj", torch.empty(B * S, 3 * Wd, device=dev, dtype=F32), M=B * S, N=3 * Wd, K=Wd, sk=sk)
            ao = ops.split_empty(B * S, Wd, sk, dev) if sk else torch.empty(B * S, Wd, device=dev, dtype=F32)
            ops.attention(qkv, qkv[:, Wd:], qkv[:, 2 * Wd:], ao, ldq=3 * Wd, ldk=3 * Wd, ldv=3 * Wd, ldo=2 * Wd if sk else Wd, n_seq=B, inner=1,
                          nq=S, nk=S, n_head=H, q_outer_stride=S, q_axis_stride=1, kv_outer_stride=S, kv_axis_stride=1,
                          kv_len=kv_len, kv_len_div=1, out_split=sk)
            _lin_fp32(ao, d, p + ".out_proj", x, M=B * S, N=Wd, K=Wd, sk=sk, a_split=ao if sk else None, residual=x, ldr=Wd)
            ops.layernorm(x, d[p + ".norm1.w"], d[p + ".norm1.b"], x, d[p + ".norm1.eps"])
            # the exact GELU is not a split-GEMM epilogue: its Linear keeps the fp32 chain in every mode
            hdn = _linear(x, d, p + ".fc1", torch.empty(B * S, 4 * Wd, device=dev, dtype=F32), F32, M=B * S, N=4 * Wd, K=Wd,
                          act=ops.ACT_GELU_ERF)
            _lin_fp32(hdn, d, p + ".fc2", x, M=B * S, N=Wd, K=4 * Wd, sk=sk, residual=x, ldr=Wd)
            ops.layernorm(x, d[p + ".norm2.w"], d[p + ".norm2.b"], x, d[p + ".norm2.eps"])
        ops.layernorm(x, d["ln_text_final.w"], d["ln_text_final.b"], x, self.ln_text_final.eps)
        out = _lin_fp32(x, d, "proj", torch.empty(B * S, self.output_dim, device=dev, dtype=F32), M=B * S, N=self.output_dim, K=Wd, sk=sk)
        return out.view(B, S, self.output_dim)


class BasicBlock(nn.Module):
    """Conv3d + GroupNorm video-prior block (mage_model.py:264-297), used by MAGE.forward with randomness=True only.
    Parameter container with the reference's keys; MAGE._video_prior runs the four blocks on the GPU (three temporal-tap
    implicit GEMMs per Conv3d + mage_groupnorm_act)."""

    def __init__(self, in_planes, out_planes, stride=1, stride_t=1, downsample=False, spectral=False):
        super().__init__()
        st = [stride_t, stride, stride]
        self.conv1 = nn.Conv3d(in_planes, out_planes, kernel_size=(3, 3, 3), stride=st, padding=1, bias=False)
        self.bn1 = nn.GroupNorm(num_groups=16, num_channels=out_planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv3d(out_planes, out_planes, kernel_size=(3, 3, 3), stride=[1, 1, 1], padding=1, bias=False)
        self.bn2 = nn.GroupNorm(num_groups=16, num_channels=out_planes)
        self.downsample = nn.Sequential(nn.Conv3d(in_planes, out_planes, kernel_size=(3, 3, 3), stride=st, padding=[1, 1, 1],
                                                  bias=False),
                                        nn.GroupNorm(num_channels=out_planes, num_groups=16)) if downsample else None
        self.stride = stride
        if spectral:
            self.conv1 = torch.nn.utils.spectral_norm(self.conv1)
            self.conv2 = torch.nn.utils.spectral_norm(self.conv2)

    forward = _no_torch_forward


class ADAIN2D(nn.Module):
    """InstanceNorm2d(x) * conv_mu(y) + conv_var(y) (mage_model.py:299-314).  forward(x, y) takes the reference's
    NCHW tensors; the GPU path runs channels-last."""

    def __init__(self, num_features, z_dim):
        super().__init__()
        self.num_features = num_features
        self.norm = nn.InstanceNorm2d(num_features, affine=False, track_running_stats=False)
        self.conv_mu = nn.Sequential(nn.Conv2d(z_dim, num_features, 3, 1, 1), nn.Conv2d(num_features, num_features, 3, 1, 1))
        self.conv_var = nn.Sequential(nn.Conv2d(z_dim, num_features, 3, 1, 1), nn.Conv2d(num_features, num_features, 3, 1, 1))
        self._derived = _Derived(self)

    def _build(self):
        d: Dict[str, torch.Tensor] = {}
        for n, seq in (("mu", self.conv_mu), ("var", self.conv_var)):
            for j in range(2):
                d[f"{n}{j}.w"] = _conv_w(seq[j])
                d[f"{n}{j}.b"] = seq[j].bias.float().contiguous()
        return d

    @torch.no_grad()
    def _run(self, x_rows: torch.Tensor, y_rows: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
        """x_rows [B*H*W, C] fp32 (motion anchor), y_rows [B*H*W, z] fp32 -> [B*H*W, C]."""
        d = self._derived.get(self._build)
        Cc, z, dev = self.num_features, y_rows.shape[1], x_rows.device
        outs = []
        for n in ("mu", "var"):
            t = VectorQuantizedVAE._conv(y_rows, d[n + "0.w"], torch.empty(B * H * W, Cc, device=dev, dtype=F32), n_img=B, H=H,
                                         W=W, cin=z, cout=Cc, k=3, bias=d[n + "0.b"])
            outs.append(VectorQuantizedVAE._conv(t, d[n + "1.w"], torch.empty(B * H * W, Cc, device=dev, dtype=F32), n_img=B,
                                                 H=H, W=W, cin=Cc, cout=Cc, k=3, bias=d[n + "1.b"]))
        return ops.adain(x_rows, outs[0], outs[1], torch.empty_like(x_rows), B=B, P=H * W, Cc=Cc, eps=self.norm.eps)

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        _need_gpu(x, "ADAIN2D")
        B, Cc, H, W = x.shape
        xr = x.permute(0, 2, 3, 1).reshape(B * H * W, Cc).float().contiguous()
        yr = y.permute(0, 2, 3, 1).reshape(B * H * W, -1).float().contiguous()
        return self._run(xr, yr, B, H, W).view(B, H, W, Cc).permute(0, 3, 1, 2)


class FlatAxialDecoder(nn.Module):
    """Axial-attention AR decoder (mage_model.py:317-390): in/context Linear -> concat along L -> + T positions ->
    `layers` blocks attending along L (causal), H, W in turn -> Linear head on x[:, 1:]."""

    def __init__(self, in_channels, model_channels, out_channels, frames_length, layers, context_channels=None, use_cids=True,
                 dropout=0.1):
        super().__init__()
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.frames_length, self.layers = frames_length, layers
        self.in_linear = nn.Linear(in_channels, model_channels)
        context_channels = default(context_channels, in_channels)
        self.context_channels = context_channels
        self.context_linear = nn.Linear(context_channels, model_channels)
        scale = model_channels ** -0.5
        self.T_positional_embedding = nn.Parameter(scale * torch.randn(frames_length, 1, 1, model_channels))
        num_heads = model_channels // 32
        self.blocks = nn.ModuleList([AxialAttentionBlock(model_channels, num_heads, axial_dim=i % 3 + 1, dropout=dropout)
                                     for i in range(layers)])
        self.use_cids = use_cids
        if use_cids:
            self.out = nn.Linear(model_channels, out_channels)
        else:
            self.out = nn.Sequential(nn.GroupNorm(32, model_channels), nn.SiLU(),
                                     zero_module(nn.Conv3d(model_channels, out_channels, 1)))
        self.initialize_parameters()
        self.compute_dtype = F32
        self.split_kind = 0                    # ops.BF16X3 / ops.F16X3: the fast parity modes (compute_dtype stays fp32)
        self.stream_bf16 = True                # bf16 mode: x stays in bf16 between the blocks (_stream_bf16)
        self._derived = _Derived(self)

    def initialize_parameters(self):
        """mage_model.py:357-365."""
        proj_std = (self.model_channels ** -0.5) * ((2 * self.layers) ** -0.5)
        attn_std = self.model_channels ** -0.5
        fc_std = (2 * self.model_channels) ** -0.5
        for block in self.blocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)

    def build_casual_attention_mask(self):
        """Kept for API parity (mage_model.py:367-372); the GPU attention kernel applies j <= i directly."""
        mask = torch.empty(self.frames_length, self.frames_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask

    def _build(self):
        d: Dict[str, torch.Tensor] = {}
        _pack_linear(d, "in_linear", self.in_linear.weight, self.in_linear.bias)
        _pack_linear(d, "context_linear", self.context_linear.weight, self.context_linear.bias)
        if self.use_cids:
            _pack_linear(d, "out", self.out.weight, self.out.bias)
        else:                                  # MAGE+ head: GroupNorm(32) + SiLU + Conv3d 1x1x1 (mage_model.py:350-354)
            d["gn.w"], d["gn.b"] = self.out[0].weight.float().contiguous(), self.out[0].bias.float().contiguous()
            w = self.out[2].weight.float().reshape(self.out_channels, self.model_channels)
            n8 = (self.out_channels + 7) // 8 * 8                        # GEMM wants N % 8 == 0: zero rows pad the 4 outputs
            wp = torch.zeros(n8, self.model_channels, device=w.device)
            wp[:self.out_channels] = w
            bp = torch.zeros(n8, device=w.device)
            bp[:self.out_channels] = self.out[2].bias.float()
            _pack_linear(d, "out", wp, bp)
        d["tpos"] = self.T_positional_embedding.float().reshape(self.frames_length, self.model_channels).contiguous()
        for i, blk in enumerate(self.blocks):
            _pack_block(d, f"b{i}", blk)
            # LayerNorm folded into the Linear that follows it (bf16 path, see _fold): W' = gamma * W per input channel,
            # s_n = sum_k W'[n, k] (of the ROUNDED weights: what the MFMA multiplies), c_n = W beta + b
            for ln, lin in (("ln_1", "in_proj"), ("ln_2", "c_fc")):
                w, g, bt = d[f"b{i}.{lin}.f32"], d[f"b{i}.{ln}.w"], d[f"b{i}.{ln}.b"]
                wq = (w * g[None, :]).to(BF16)
                d[f"b{i}.{lin}.lnw"] = wq                   # (f16 mode: `.lnw.f16` / `.lns.f16`, built on first use by _ln_fold_w)
                d[f"b{i}.{lin}.lns"] = wq.float().sum(dim=1).contiguous()
                # c_n = sum_k W_nk beta_k + b_n in fp64 (a row-wise reduction: derived-cache bookkeeping, no BLAS call)
                d[f"b{i}.{lin}.lnc"] = ((w.double() * bt.double()[None, :]).sum(1) + d[f"b{i}.{lin}.b"].double()).float().contiguous()
        return d

    def _fold(self, dt, B: int, hw: int) -> bool:
        """bf16 path: the standalone LayerNorm launches between the GEMMs are folded into them (csrc/gemm.hip, epilogue_lean):
        the x + Linear(.) GEMM also writes a bf16 copy of x and per-row partial (sum, sum of squares); the next Linear takes that
        copy with gamma folded into its weights and finishes the normalisation in its epilogue.  Needs whole 256-row tiles per
        frame slot (so that the full pass and the incremental loop take the same route: their tokens stay bit-identical)."""
        return (dt in HALF_TYPES and (B * hw) % 256 == 0 and self.model_channels % 256 == 0 and config.get().ln_fold)

    def _stream_bf16(self) -> bool:
        """bf16 mode with the LayerNorm fold: x itself stays in bf16 between the blocks -- every x + Linear(.) reads the bf16 rows as its
        residual and writes bf16 rows (+ the LayerNorm partial sums of the fp32 values before rounding); no fp32 stream, no second copy:
        -0.8 GB of HBM traffic per x + Linear(.) launch at cfg2.  Measured against the fp32-stream form (`stream_bf16 = False` or
        config stream_16bit = False (MAGE_STREAM_FP32=1)): first-generated-frame token agreement with the fp32-class modes 0.956 vs 0.958 (16 clips), i.e. inside the
        bf16 GEMM noise.  The incremental loop uses the same kernels on the same rows: still bit-identical to the full loop."""
        return getattr(self, "stream_bf16", True) and config.get().stream_16bit

    def _ln_linear(self, d, p, lin, xb, stats, y, *, M, N, lo=0, hi=None, part=None, **kw):
        """y = Linear(LN(x)) from the bf16 copy of x and its row statistics (rows lo:hi of the Linear's outputs).  part (instead of
        stats): the producer's partial sums -- few-rows GEMMs reduce them in their prologue (_stats_inline), no mage_ln_stats launch."""
        Cc = self.model_channels
        hi = N + lo if hi is None else hi
        src = dict(ln_part=part, ln_eps=1e-5) if part is not None else dict(ln_stats=stats)
        lnw, lns = self._ln_fold_w(d, p, lin, xb.dtype)
        return ops.gemm(xb, lnw[lo:hi], y, M=M, N=hi - lo, K=Cc, lda=Cc, ldy=kw.pop("ldy", hi - lo),
                        bias=d[f"{p}.{lin}.lnc"][lo:hi], ln_colsum=lns[lo:hi], **src, **kw)

    @staticmethod
    def _ln_fold_w(d, p, lin, dt):
        """(W' = gamma * W rounded to dt, s_n = the row sums of the ROUNDED W') of the LayerNorm fold (see _build): bf16 from _build, f16 on
        first use."""
        if dt == BF16:
            return d[f"{p}.{lin}.lnw"], d[f"{p}.{lin}.lns"]
        kw_, ks_ = f"{p}.{lin}.lnw.f16", f"{p}.{lin}.lns.f16"
        if kw_ not in d:
            ln = "ln_1" if lin == "in_proj" else "ln_2"
            wq = (d[f"{p}.{lin}.f32"] * d[f"{p}.{ln}.w"][None, :]).to(F16)
            d[kw_], d[ks_] = wq, wq.float().sum(dim=1).contiguous()
        return d[kw_], d[ks_]

    def _stats_inline(self, xb, M: int) -> bool:
        """One clip per call: every Linear that follows a LayerNorm (N = C .. 4C) runs on the few-rows kernel, which reduces the
        producer's partial sums itself (same arithmetic as mage_ln_stats, mage_ln_stats_row in csrc/common.h).  (The tiled kernels doing the
        same at the incremental loop's 16 k rows was measured in round 5: slower than the 7 us launch it removes.)"""
        Cc = self.model_channels
        return all(ops.gemm_is_small(xb, M, n, Cc) for n in (Cc, 2 * Cc, 3 * Cc, 4 * Cc))

    @torch.no_grad()
    def _run(self, motion: torch.Tensor, imgs: torch.Tensor, *, B: int, hh: int, ww: int) -> torch.Tensor:
        """motion [B*hw, Cc], imgs [B*(L-1)*hw, Ci] in the compute dtype -> logits [B*(L-1)*hw, K] fp32
        (use_cids=False: predicted latents [B*(L-1)*hw, 8] fp32 whose first out_channels columns are valid)."""
        if self._split_on():
            return self._run_split(motion, imgs, B=B, hh=hh, ww=ww)
        d = self._derived.get(self._build)
        dt, Cc, L, dev = self.compute_dtype, self.model_channels, self.frames_length, motion.device
        hw = hh * ww
        M = B * L * hw
        H = Cc // 32
        fold, have_stats = self._fold(dt, B, hw), False
        sb = fold and self._stream_bf16()
        inl = fill_stats = False
        if fold:
            xb = torch.empty(M, Cc, device=dev, dtype=dt)                          # the bf16 stream (or the bf16 copy of the fp32 one)
            part = torch.empty(Cc // 64, M, 2, device=dev, dtype=F32)       # slice-major (mage_gemm_desc::ln_part_rows)
            stats = torch.empty(M, 2, device=dev, dtype=F32)
            inl = self._stats_inline(xb, M)
        # the stream starts in bf16 too: context_linear and the frame fill write bf16 rows, their LayerNorm statistics come from one pass
        # over those rows (mage_row_stats), block 0 then runs like every other block (no fp32 rows, no LayerNorm launch)
        x0 = xb if sb else torch.empty(M, Cc, device=dev, dtype=F32)               # residual stream as assembled by :375-378
        x = x0 if not sb else None
        # context_linear -> slot 0, in_linear -> slots 1..L-1, + T_positional_embedding, no concat copy (:375-378)
        _linear(motion, d, "context_linear", x0, dt, M=B * hw, N=Cc, K=self.context_channels, out_w=hw, y_img_stride=L * hw,
                rowadd=d["tpos"], rowadd_div=hw, rowadd_mod=L)
        if isinstance(imgs, FrameTokens):
            imgs.fill(x0, n_img=B * (L - 1), per_clip=L - 1, P=L, y_off=hw, tp=d["tpos"])
        else:
            _linear(imgs, d, "in_linear", x0, dt, M=B * (L - 1) * hw, N=Cc, K=self.in_channels, out_w=(L - 1) * hw,
                    y_img_stride=L * hw, y_off=hw, rowadd=d["tpos"], rowadd_div=hw, rowadd_mod=L)
        if sb:
            ops.row_stats(xb, 1e-5, stats)
            have_stats = fill_stats = True
        xn = torch.empty(M, Cc, device=dev, dtype=dt)
        qkv = torch.empty(M, 3 * Cc, device=dev, dtype=dt)
        ao = torch.empty(M, Cc, device=dev, dtype=dt)
        hdn = torch.empty(M, 4 * Cc, device=dev, dtype=dt)
        for i in range(self.layers):
            p = f"b{i}"
            axis = i % 3                                                            # 0: L (causal), 1: H, 2: W  (:344,:382)
            if axis == 0:
                geo = dict(n_seq=B * hw, inner=hw, nq=L, nk=L, q_outer_stride=L * hw, q_axis_stride=hw, causal=True)
            elif axis == 1:
                geo = dict(n_seq=B * L * ww, inner=ww, nq=hh, nk=hh, q_outer_stride=hw, q_axis_stride=ww, causal=False)
            else:
                geo = dict(n_seq=B * L * hh, inner=1, nq=ww, nk=ww, q_outer_stride=ww, q_axis_stride=1, causal=False)
            if have_stats:
                self._ln_linear(d, p, "in_proj", xb, stats, qkv, M=M, N=3 * Cc, part=part if (inl and not (fill_stats and i == 0)) else None)
            else:
                ops.layernorm(x, d[p + ".ln_1.w"], d[p + ".ln_1.b"], xn, 1e-5)
                _linear(xn, d, p + ".in_proj", qkv, dt, M=M, N=3 * Cc, K=Cc)
            ops.attention(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], ao, ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc, ldo=Cc, n_head=H,
                          kv_outer_stride=geo["q_outer_stride"], kv_axis_stride=geo["q_axis_stride"], **geo)
            if fold:
                if sb:
                    _linear(ao, d, p + ".out_proj", xb, dt, M=M, N=Cc, K=Cc, residual=xb, ldr=Cc, ln_part=part)
                else:
                    _linear(ao, d, p + ".out_proj", x, dt, M=M, N=Cc, K=Cc, residual=x, ldr=Cc, y2=xb, ldy2=Cc, ln_part=part)
                if not inl:
                    ops.ln_stats(part, Cc, 1e-5, stats)
                self._ln_linear(d, p, "c_fc", xb, stats, hdn, M=M, N=4 * Cc, act=ops.ACT_QUICKGELU, part=part if inl else None)
            else:
                _linear(ao, d, p + ".out_proj", x, dt, M=M, N=Cc, K=Cc, residual=x, ldr=Cc)
                ops.layernorm(x, d[p + ".ln_2.w"], d[p + ".ln_2.b"], xn, 1e-5)
                _linear(xn, d, p + ".c_fc", hdn, dt, M=M, N=4 * Cc, K=Cc, act=ops.ACT_QUICKGELU)
            last = i == self.layers - 1
            if last and self.use_cids and dt != F32:
                # the last block's x + c_proj(.) is only read by the head GEMM: the epilogue rounds it to the compute dtype on the
                # way out (same fp32 sum, same round-to-nearest-even as a separate cast pass: bit-identical) instead of writing
                # the fp32 stream and converting it in another launch
                _linear(hdn, d, p + ".c_proj", xn, dt, M=M, N=Cc, K=4 * Cc, residual=xb if sb else x, ldr=Cc)
            elif fold and not last:
                if sb:
                    _linear(hdn, d, p + ".c_proj", xb, dt, M=M, N=Cc, K=4 * Cc, residual=xb, ldr=Cc, ln_part=part)
                else:
                    _linear(hdn, d, p + ".c_proj", x, dt, M=M, N=Cc, K=4 * Cc, residual=x, ldr=Cc, y2=xb, ldy2=Cc, ln_part=part)
                if not inl:
                    ops.ln_stats(part, Cc, 1e-5, stats)
                have_stats = True
            else:
                if x is None:                                            # bf16 stream, MAGE+ head: its GroupNorm reads fp32 rows
                    x = torch.empty(M, Cc, device=dev, dtype=F32)
                _linear(hdn, d, p + ".c_proj", x, dt, M=M, N=Cc, K=4 * Cc, residual=xb if sb else x, ldr=Cc)
        if not self.use_cids:
            # GroupNorm statistics span all L-1 frames of a clip (:387-388): this head is NOT causal along L
            y = ops.groupnorm_silu(x, d["gn.w"], d["gn.b"], torch.empty(B * (L - 1) * hw, Cc, device=dev, dtype=dt), n_samples=B,
                                   rows_per_sample=(L - 1) * hw, sample_stride_rows=L * hw, row_off=hw, groups=32,
                                   eps=self.out[0].eps)
            n8 = d["out.f32"].shape[0]
            pred = torch.empty(B * (L - 1) * hw, n8, device=dev, dtype=F32)
            return _linear(y, d, "out", pred, dt, M=B * (L - 1) * hw, N=n8, K=Cc)
        xa = x if dt == F32 else xn
        logits = torch.empty(B * (L - 1) * hw, self.out_channels, device=dev, dtype=F32)
        _linear(xa, d, "out", logits, dt, M=B * (L - 1) * hw, N=self.out_channels, K=Cc, out_w=(L - 1) * hw,
                a_img_stride=L * hw, a_off=hw)                                      # head on x[:, 1:]  (:385)
        return logits

    # ------------------------------------------------------------------ incremental (temporal KV cache) decoding
    # the decoder is causal along L (temporal blocks masked, spatial blocks per frame), so the logits of
    # frame k do not depend on later slots.  Instead of recomputing all L positions in each of the L-1 iterations
    # (mage_model.py:673-684) only the NEW position(s) run through the stack; the temporal blocks keep their K,V of
    # earlier positions in a cache.  Same kernels, same per-row arithmetic: the tokens are bit-identical to the full loop.
    @torch.no_grad()
    def _inc_begin(self, B: int, hh: int, ww: int, device) -> dict:
        dt, Cc, L = self.compute_dtype, self.model_channels, self.frames_length
        if self._split_on() and self._attn_split():        # f16x3: K, V cached as split rows (what the attention kernel reads)
            caches = {i: ops.split_empty(B * L * hh * ww, 2 * Cc, self.split_kind, device) for i in range(self.layers) if i % 3 == 0}
            return {"B": B, "hh": hh, "ww": ww, "kv": caches, "p": 0}
        if self._split_on():                                # the other split forms (_inc_step_split with fp32 rows): [K | V] only, q stays in qkv
            caches = {i: torch.empty(B * L * hh * ww, 2 * Cc, device=device, dtype=dt) for i in range(self.layers) if i % 3 == 0}
            return {"B": B, "hh": hh, "ww": ww, "kv": caches, "p": 0}
        # temporal blocks: one row [q | k | v] per (clip, slot, pixel) -- the new positions' QKV projection is ONE launch writing straight into
        # the slots (q is read back from there by the attention of the same step; 1.5x the K,V bytes of a cache that is 0.8 GB at cfg2)
        caches = {i: torch.empty(B * L * hh * ww, 3 * Cc, device=device, dtype=dt) for i in range(self.layers) if i % 3 == 0}
        return {"B": B, "hh": hh, "ww": ww, "kv": caches, "p": 0}

    @torch.no_grad()
    def _inc_step(self, st: dict, motion: Optional[torch.Tensor], imgs: torch.Tensor) -> torch.Tensor:
        """Append position(s): the first call takes the motion anchor (slot 0) and frame 0's features (slot 1); later calls
        take the features of the newest frame only.  Returns the logits of the last appended slot, [B*hw, K] fp32."""
        if self._split_on():
            return self._inc_step_split(st, motion, imgs)
        d = self._derived.get(self._build)
        dt, Cc, L = self.compute_dtype, self.model_channels, self.frames_length
        dev = imgs.tokens.device if isinstance(imgs, FrameTokens) else imgs.device
        B, hh, ww = st["B"], st["hh"], st["ww"]
        hw, H = hh * ww, Cc // 32
        p0 = st["p"]
        P = 2 if motion is not None else 1                                  # new positions p0 .. p0+P-1
        assert (p0 == 0) == (motion is not None) and p0 + P <= L
        M = B * P * hw
        fold, have_stats = self._fold(dt, B, hw), False
        sb = fold and self._stream_bf16()
        inl = fill_stats = False
        if fold:                                                             # see _run
            xb = torch.empty(M, Cc, device=dev, dtype=dt)
            part = torch.empty(Cc // 64, M, 2, device=dev, dtype=F32)       # slice-major (mage_gemm_desc::ln_part_rows)
            stats = torch.empty(M, 2, device=dev, dtype=F32)
            inl = self._stats_inline(xb, M)
        x0 = xb if sb else torch.empty(M, Cc, device=dev, dtype=F32)
        x = x0 if not sb else None
        tp = d["tpos"][p0:]
        if motion is not None:
            _linear(motion, d, "context_linear", x0, dt, M=B * hw, N=Cc, K=self.context_channels, out_w=hw, y_img_stride=P * hw,
                    rowadd=tp, rowadd_div=hw, rowadd_mod=P)
        if isinstance(imgs, FrameTokens):
            imgs.fill(x0, n_img=B, per_clip=1, P=P, y_off=(P - 1) * hw, tp=tp)
        else:
            _linear(imgs, d, "in_linear", x0, dt, M=B * hw, N=Cc, K=self.in_channels, out_w=hw, y_img_stride=P * hw,
                    y_off=(P - 1) * hw, rowadd=tp, rowadd_div=hw, rowadd_mod=P)
        if sb:
            ops.row_stats(xb, 1e-5, stats)
            have_stats = fill_stats = True
        xn = torch.empty(M, Cc, device=dev, dtype=dt)
        qkv = torch.empty(M, 3 * Cc, device=dev, dtype=dt)
        ao = torch.empty(M, Cc, device=dev, dtype=dt)
        hdn = torch.empty(M, 4 * Cc, device=dev, dtype=dt)
        for i in range(self.layers):
            p = f"b{i}"
            axis = i % 3
            if not have_stats:
                ops.layernorm(x, d[p + ".ln_1.w"], d[p + ".ln_1.b"], xn, 1e-5)
            w, b = _wdt(d, p + ".in_proj", dt), d[p + ".in_proj.b"]
            pin = part if (inl and not (fill_stats and i == 0)) else None        # block 0: the statistics of the fill, not partial sums
            if axis == 0:
                kv = st["kv"][i]                                             # [B, L, hw, Q|K|V]
                if have_stats:                                               # q, k, v of the new positions -> their cache slots, one launch
                    self._ln_linear(d, p, "in_proj", xb, stats, kv, M=M, N=3 * Cc, out_w=P * hw, y_img_stride=L * hw, y_off=p0 * hw, part=pin)
                else:
                    ops.gemm(xn, w, kv, M=M, N=3 * Cc, K=Cc, lda=Cc, ldy=3 * Cc, bias=b, out_w=P * hw, y_img_stride=L * hw, y_off=p0 * hw)
                ops.attention(kv[p0 * hw:], kv[:, Cc:], kv[:, 2 * Cc:], ao, ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc, ldo=Cc, n_seq=B * hw, inner=hw,
                              nq=P, nk=p0 + P, n_head=H, q_outer_stride=L * hw, q_axis_stride=hw, kv_outer_stride=L * hw,
                              kv_axis_stride=hw, causal=True, o_outer_stride=P * hw, o_axis_stride=hw)
            else:
                if have_stats:
                    self._ln_linear(d, p, "in_proj", xb, stats, qkv, M=M, N=3 * Cc, part=pin)
                else:
                    ops.gemm(xn, w, qkv, M=M, N=3 * Cc, K=Cc, lda=Cc, ldy=3 * Cc, bias=b)
                if axis == 1:
                    geo = dict(n_seq=B * P * ww, inner=ww, nq=hh, nk=hh, q_outer_stride=hw, q_axis_stride=ww)
                else:
                    geo = dict(n_seq=B * P * hh, inner=1, nq=ww, nk=ww, q_outer_stride=ww, q_axis_stride=1)
                ops.attention(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], ao, ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc, ldo=Cc, n_head=H,
                              kv_outer_stride=geo["q_outer_stride"], kv_axis_stride=geo["q_axis_stride"], **geo)
            if fold:
                if sb:
                    _linear(ao, d, p + ".out_proj", xb, dt, M=M, N=Cc, K=Cc, residual=xb, ldr=Cc, ln_part=part)
                else:
                    _linear(ao, d, p + ".out_proj", x, dt, M=M, N=Cc, K=Cc, residual=x, ldr=Cc, y2=xb, ldy2=Cc, ln_part=part)
                if not inl:
                    ops.ln_stats(part, Cc, 1e-5, stats)
                self._ln_linear(d, p, "c_fc", xb, stats, hdn, M=M, N=4 * Cc, act=ops.ACT_QUICKGELU, part=part if inl else None)
            else:
                _linear(ao, d, p + ".out_proj", x, dt, M=M, N=Cc, K=Cc, residual=x, ldr=Cc)
                ops.layernorm(x, d[p + ".ln_2.w"], d[p + ".ln_2.b"], xn, 1e-5)
                _linear(xn, d, p + ".c_fc", hdn, dt, M=M, N=4 * Cc, K=Cc, act=ops.ACT_QUICKGELU)
            last = i == self.layers - 1
            if last and dt != F32:                                  # see _run: the head's input straight from the epilogue
                _linear(hdn, d, p + ".c_proj", xn, dt, M=M, N=Cc, K=4 * Cc, residual=xb if sb else x, ldr=Cc)
            elif fold and not last:
                if sb:
                    _linear(hdn, d, p + ".c_proj", xb, dt, M=M, N=Cc, K=4 * Cc, residual=xb, ldr=Cc, ln_part=part)
                else:
                    _linear(hdn, d, p + ".c_proj", x, dt, M=M, N=Cc, K=4 * Cc, residual=x, ldr=Cc, y2=xb, ldy2=Cc, ln_part=part)
                if not inl:
                    ops.ln_stats(part, Cc, 1e-5, stats)
                have_stats = True
            else:
                _linear(hdn, d, p + ".c_proj", x, dt, M=M, N=Cc, K=4 * Cc, residual=xb if sb else x, ldr=Cc)
        xa = x if dt == F32 else xn
        logits = torch.empty(B * hw, self.out_channels, device=dev, dtype=F32)
        _linear(xa, d, "out", logits, dt, M=B * hw, N=self.out_channels, K=Cc, out_w=hw, a_img_stride=P * hw, a_off=(P - 1) * hw)
        st["p"] = p0 + P
        return logits


    # ------------------------------------------------------------------ the fast parity modes ('f16x3' / 'bf16x3')
    # The stack of _run / _inc_step with fp32 everywhere EXCEPT the operands of the Linear layers: those are split-precision
    # tensors (two 16-bit pieces per element, ops.split_empty) written directly by their producers -- LayerNorm, the attention
    # kernel, the c_fc epilogue (QuickGELU), the last c_proj epilogue -- and multiplied as three f16 / bf16 MFMA products per K
    # slab (include/mage_hip.h, MAGE_F16X3).  Residual stream, q / k / v, softmax and logits are fp32 as in 'fp32' mode.
    def _split_on(self) -> bool:
        return bool(self.split_kind) and self.compute_dtype == F32 and self.model_channels % 64 == 0

    def _warm_split(self) -> None:
        """Build every split-precision weight copy now (MAGE._warm_derived: on the caller's stream, before side streams fork)."""
        if not self._split_on():
            return
        d = self._derived.get(self._build)
        names = ["in_linear", "context_linear"] + (["out"] if self.use_cids else [])
        for i in range(self.layers):
            names += [f"b{i}.in_proj", f"b{i}.out_proj", f"b{i}.c_fc", f"b{i}.c_proj"]
        for n in names:
            _wsplit(d, n, self.split_kind)

    def _attn_split(self) -> bool:
        """f16x3: the axial attentions read split q, k, v on the matrix cores (attention_mfma_split_kernel); sequences up to 32."""
        return (self.split_kind == ops.F16X3 and self.frames_length <= 32 and (self.model_channels // 32) % 2 == 0
                and config.get().attn_split)

    def _taps_ok(self, rows: int) -> bool:
        """in_linear / context_linear (+ T positions) and the frame convolution run as the padded-taps form of the split GEMM."""
        return self.model_channels % 256 == 0 and rows % 256 == 0 and self.in_channels % 64 == 0 and self.context_channels % 64 == 0

    def _lin_s(self, a, d, name, y, *, M, N, K, lo=0, hi=None, **kw):
        sk = self.split_kind
        w, b = _wsplit(d, name, sk), d.get(name + ".b")
        if hi is not None or lo:
            w, b = w[lo:hi], (None if b is None else b[lo:hi])
        y_split = kw.pop("y_split", False)
        return ops.gemm(a, w, y, M=M, N=N, K=K, lda=kw.pop("lda", 2 * K), ldy=kw.pop("ldy", 2 * N if y_split else N), bias=b,
                        split_kind=sk, y_split=y_split, **kw)

    def _embed_inputs_split(self, d, x, motion, imgs, *, B, hw, P, tp, n_img_rows):
        """context_linear -> slot 0 (if motion), in_linear -> the other slot(s), + T positions: rows of x [B*P*hw, C] fp32.
        motion fp32 rows; imgs split rows (from _frame_features in split mode) or fp32 rows."""
        sk, Cc = self.split_kind, self.model_channels
        img_tok = isinstance(imgs, FrameTokens)
        img_split = (not img_tok) and imgs.dtype != F32
        if motion is not None:
            if self._taps_ok(B * hw):
                self._lin_s(ops.split(motion, sk), d, "context_linear", x, M=B * hw, N=Cc, K=self.context_channels, out_w=hw,
                            y_img_stride=P * hw, rowadd=tp, rowadd_div=hw, rowadd_mod=P)
            else:
                _linear(motion, d, "context_linear", x, F32, M=B * hw, N=Cc, K=self.context_channels, out_w=hw, y_img_stride=P * hw,
                        rowadd=tp, rowadd_div=hw, rowadd_mod=P)
        off = hw if motion is not None else 0
        if img_tok:
            imgs.fill(x, n_img=n_img_rows // hw, per_clip=n_img_rows // hw // B, P=P, y_off=off, tp=tp)
        elif img_split:
            self._lin_s(imgs, d, "in_linear", x, M=n_img_rows, N=Cc, K=self.in_channels, out_w=n_img_rows // B, y_img_stride=P * hw,
                        y_off=off, rowadd=tp, rowadd_div=hw, rowadd_mod=P)
        else:
            _linear(imgs, d, "in_linear", x, F32, M=n_img_rows, N=Cc, K=self.in_channels, out_w=n_img_rows // B, y_img_stride=P * hw,
                    y_off=off, rowadd=tp, rowadd_div=hw, rowadd_mod=P)

    @torch.no_grad()
    def _run_split(self, motion: torch.Tensor, imgs: torch.Tensor, *, B: int, hh: int, ww: int) -> torch.Tensor:
        d = self._derived.get(self._build)
        sk, Cc, L, dev = self.split_kind, self.model_channels, self.frames_length, motion.device
        hw = hh * ww
        M = B * L * hw
        H = Cc // 32
        x = torch.empty(M, Cc, device=dev, dtype=F32)
        self._embed_inputs_split(d, x, motion, imgs, B=B, hw=hw, P=L, tp=d["tpos"], n_img_rows=B * (L - 1) * hw)
        xn = ops.split_empty(M, Cc, sk, dev)
        qs = self._attn_split()                # f16x3: q, k, v leave the QKV epilogue as split rows, attention on the matrix cores
        qkv = ops.split_empty(M, 3 * Cc, sk, dev) if qs else torch.empty(M, 3 * Cc, device=dev, dtype=F32)
        ao = ops.split_empty(M, Cc, sk, dev)
        hdn = ops.split_empty(M, 4 * Cc, sk, dev)
        for i in range(self.layers):
            p = f"b{i}"
            axis = i % 3
            if axis == 0:
                geo = dict(n_seq=B * hw, inner=hw, nq=L, nk=L, q_outer_stride=L * hw, q_axis_stride=hw, causal=True)
            elif axis == 1:
                geo = dict(n_seq=B * L * ww, inner=ww, nq=hh, nk=hh, q_outer_stride=hw, q_axis_stride=ww, causal=False)
            else:
                geo = dict(n_seq=B * L * hh, inner=1, nq=ww, nk=ww, q_outer_stride=ww, q_axis_stride=1, causal=False)
            ops.layernorm(x, d[p + ".ln_1.w"], d[p + ".ln_1.b"], xn, 1e-5, split_kind=sk)
            self._lin_s(xn, d, p + ".in_proj", qkv, M=M, N=3 * Cc, K=Cc, y_split=qs)
            m_ = 2 if qs else 1                                                      # split rows: 2 16-bit elements per logical column
            ops.attention(qkv, qkv[:, m_ * Cc:], qkv[:, m_ * 2 * Cc:], ao, ldq=m_ * 3 * Cc, ldk=m_ * 3 * Cc, ldv=m_ * 3 * Cc, ldo=2 * Cc, n_head=H,
                          kv_outer_stride=geo["q_outer_stride"], kv_axis_stride=geo["q_axis_stride"], out_split=sk, split_kind=sk if qs else 0,
                          **geo)
            self._lin_s(ao, d, p + ".out_proj", x, M=M, N=Cc, K=Cc, residual=x, ldr=Cc)
            ops.layernorm(x, d[p + ".ln_2.w"], d[p + ".ln_2.b"], xn, 1e-5, split_kind=sk)
            self._lin_s(xn, d, p + ".c_fc", hdn, M=M, N=4 * Cc, K=Cc, act=ops.ACT_QUICKGELU, y_split=True)
            if i == self.layers - 1 and self.use_cids:       # only the head reads the last x: it leaves the epilogue as split rows
                self._lin_s(hdn, d, p + ".c_proj", xn, M=M, N=Cc, K=4 * Cc, residual=x, ldr=Cc, y_split=True)
            else:
                self._lin_s(hdn, d, p + ".c_proj", x, M=M, N=Cc, K=4 * Cc, residual=x, ldr=Cc)
        if not self.use_cids:                                 # MAGE+ head (N = 8 columns): GroupNorm + SiLU, fp32 GEMM
            y = ops.groupnorm_silu(x, d["gn.w"], d["gn.b"], torch.empty(B * (L - 1) * hw, Cc, device=dev, dtype=F32), n_samples=B,
                                   rows_per_sample=(L - 1) * hw, sample_stride_rows=L * hw, row_off=hw, groups=32,
                                   eps=self.out[0].eps)
            n8 = d["out.f32"].shape[0]
            pred = torch.empty(B * (L - 1) * hw, n8, device=dev, dtype=F32)
            return _linear(y, d, "out", pred, F32, M=B * (L - 1) * hw, N=n8, K=Cc)
        logits = torch.empty(B * (L - 1) * hw, self.out_channels, device=dev, dtype=F32)
        self._lin_s(xn, d, "out", logits, M=B * (L - 1) * hw, N=self.out_channels, K=Cc, out_w=(L - 1) * hw, a_img_stride=L * hw,
                    a_off=hw)
        return logits

    @torch.no_grad()
    def _inc_step_split(self, st: dict, motion: Optional[torch.Tensor], imgs: torch.Tensor) -> torch.Tensor:
        d = self._derived.get(self._build)
        sk, Cc, L = self.split_kind, self.model_channels, self.frames_length
        dev = imgs.tokens.device if isinstance(imgs, FrameTokens) else imgs.device
        B, hh, ww = st["B"], st["hh"], st["ww"]
        hw, H = hh * ww, Cc // 32
        p0 = st["p"]
        P = 2 if motion is not None else 1
        assert (p0 == 0) == (motion is not None) and p0 + P <= L
        M = B * P * hw
        x = torch.empty(M, Cc, device=dev, dtype=F32)
        self._embed_inputs_split(d, x, motion, imgs, B=B, hw=hw, P=P, tp=d["tpos"][p0:], n_img_rows=B * hw)
        xn = ops.split_empty(M, Cc, sk, dev)
        qs = self._attn_split()
        m_ = 2 if qs else 1                                                  # split rows: 2 16-bit elements per logical column
        ask = sk if qs else 0
        qkv = ops.split_empty(M, 3 * Cc, sk, dev) if qs else torch.empty(M, 3 * Cc, device=dev, dtype=F32)
        ao = ops.split_empty(M, Cc, sk, dev)
        hdn = ops.split_empty(M, 4 * Cc, sk, dev)
        for i in range(self.layers):
            p = f"b{i}"
            axis = i % 3
            ops.layernorm(x, d[p + ".ln_1.w"], d[p + ".ln_1.b"], xn, 1e-5, split_kind=sk)
            if axis == 0:
                kv = st["kv"][i]                                             # [B, L, hw, K|V] fp32 (f16x3: split rows)
                qv = qkv.view(-1)[:M * m_ * Cc].view(M, m_ * Cc)             # q of the new positions, packed [M, C]
                self._lin_s(xn, d, p + ".in_proj", qv, M=M, N=Cc, K=Cc, lo=0, hi=Cc, ldy=m_ * Cc, y_split=qs)
                self._lin_s(xn, d, p + ".in_proj", kv, M=M, N=2 * Cc, K=Cc, lo=Cc, hi=3 * Cc, ldy=m_ * 2 * Cc, out_w=P * hw,
                            y_img_stride=L * hw, y_off=p0 * hw, y_split=qs)
                ops.attention(qv, kv, kv[:, m_ * Cc:], ao, ldq=m_ * Cc, ldk=m_ * 2 * Cc, ldv=m_ * 2 * Cc, ldo=2 * Cc, n_seq=B * hw, inner=hw, nq=P,
                              nk=p0 + P, n_head=H, q_outer_stride=P * hw, q_axis_stride=hw, kv_outer_stride=L * hw,
                              kv_axis_stride=hw, causal=True, out_split=sk, split_kind=ask)
            else:
                self._lin_s(xn, d, p + ".in_proj", qkv, M=M, N=3 * Cc, K=Cc, y_split=qs)
                if axis == 1:
                    geo = dict(n_seq=B * P * ww, inner=ww, nq=hh, nk=hh, q_outer_stride=hw, q_axis_stride=ww)
                else:
                    geo = dict(n_seq=B * P * hh, inner=1, nq=ww, nk=ww, q_outer_stride=ww, q_axis_stride=1)
                ops.attention(qkv, qkv[:, m_ * Cc:], qkv[:, m_ * 2 * Cc:], ao, ldq=m_ * 3 * Cc, ldk=m_ * 3 * Cc, ldv=m_ * 3 * Cc, ldo=2 * Cc, n_head=H,
                              kv_outer_stride=geo["q_outer_stride"], kv_axis_stride=geo["q_axis_stride"], out_split=sk, split_kind=ask, **geo)
            self._lin_s(ao, d, p + ".out_proj", x, M=M, N=Cc, K=Cc, residual=x, ldr=Cc)
            ops.layernorm(x, d[p + ".ln_2.w"], d[p + ".ln_2.b"], xn, 1e-5, split_kind=sk)
            self._lin_s(xn, d, p + ".c_fc", hdn, M=M, N=4 * Cc, K=Cc, act=ops.ACT_QUICKGELU, y_split=True)
            if i == self.layers - 1:
                self._lin_s(hdn, d, p + ".c_proj", xn, M=M, N=Cc, K=4 * Cc, residual=x, ldr=Cc, y_split=True)
            else:
                self._lin_s(hdn, d, p + ".c_proj", x, M=M, N=Cc, K=4 * Cc, residual=x, ldr=Cc)
        logits = torch.empty(B * hw, self.out_channels, device=dev, dtype=F32)
        self._lin_s(xn, d, "out", logits, M=B * hw, N=self.out_channels, K=Cc, out_w=hw, a_img_stride=P * hw, a_off=(P - 1) * hw)
        st["p"] = p0 + P
        return logits

    def forward(self, motion: torch.Tensor, imgs: torch.Tensor) -> torch.Tensor:
        """motion [B,H,W,Cc], imgs [B,L-1,H,W,Ci] -> logits [B,L-1,H,W,out] fp32."""
        _need_gpu(motion, "FlatAxialDecoder")
        B, hh, ww, _ = motion.shape
        dt = self.compute_dtype
        m = motion.reshape(B * hh * ww, -1).to(dt).contiguous()
        im = imgs.reshape(-1, imgs.shape[-1]).to(dt).contiguous()
        out = self._run(m, im, B=B, hh=hh, ww=ww)
        return out.view(B, self.frames_length - 1, hh, ww, -1)[..., :self.out_channels]


class PIDControl:
    """KL-weight PI controller (mage_model.py:394-434); host-side scalar bookkeeping, training only."""

    def __init__(self):
        self.I_k1 = 0.0
        self.W_k1 = 0.0
        self.e_k1 = 0.0

    def _Kp_fun(self, Err, scale=1):
        return 1.0 / (1.0 + float(scale) * exp(Err))

    def pid(self, exp_KL, KL_loss, Kp=0.01, Ki=-0.0001, Kd=0.0):
        error_k = exp_KL - KL_loss
        Pk = Kp * self._Kp_fun(error_k)
        Ik = self.I_k1 + Ki * error_k
        if self.W_k1 < 0 and self.W_k1 >= 1:      # (sic) never true in the reference either
            Ik = self.I_k1
        Wk = Pk + Ik
        self.W_k1, self.I_k1, self.e_k1 = Wk, Ik, error_k
        return min(max(Wk, 0.0), 1.0), error_k


def disabled_train(self, mode=True):
    return self


class MAGE(nn.Module):
    def __init__(self, first_stage_config, text_encoder_config, ma_config, generate_decoder_config, codebook_size: int,
                 frames_length: int, image_resolution: int, vision_width: int, dropout: float = 0.1, use_cids=False,
                 randomness=False, alpha=0., beta=1., v_kl=0., auto_beta=False):
        super().__init__()
        self.instantiate_first_stage(first_stage_config)
        self.frames_length, self.image_resolution, self.vision_width = frames_length, image_resolution, vision_width
        self.dropout, self.use_cids, self.auto_beta = dropout, use_cids, auto_beta
        self.text_encoder = instantiate_from_config(text_encoder_config)
        self.ma_encoder = instantiate_from_config(ma_config, {"dropout": dropout})
        # The reference ships ONE TransformerBlock.forward with two alternative first lines (mage_model.py:92-93) and tells the
        # user to swap them by hand for MAGE+.  MAGE+ is exactly the use_cids=False family (config/mage+_*.yaml), so the variant
        # follows that switch; `model.ma_encoder.mage_plus = False` restores the shipped line for a MAGE+ config.
        if hasattr(self.ma_encoder, "mage_plus"):
            self.ma_encoder.mage_plus = not use_cids
        self.generate_model = instantiate_from_config(
            generate_decoder_config, {"use_cids": use_cids, "dropout": dropout, "context_channels": ma_config["params"]["d_model"]})
        self.codebook_size = codebook_size
        if use_cids:
            self.visual_token_embedding = nn.Embedding(codebook_size, vision_width)
        else:
            self.visual_token_embedding = nn.Linear(self.first_stage_model.embed_dim, vision_width)
        self.conv = nn.Sequential(nn.Conv2d(vision_width, vision_width, kernel_size=3, stride=1, padding=1, bias=False))
        scale = vision_width ** -0.5
        self.speed_embedding = nn.Parameter(scale * torch.randn(1, vision_width))
        self.H_positional_embedding = nn.Parameter(scale * torch.randn(1, image_resolution, 1, vision_width))
        self.W_positional_embedding = nn.Parameter(scale * torch.randn(1, 1, image_resolution, vision_width))
        self.randomness = randomness
        if randomness:
            dm = ma_config["params"]["d_model"]
            self.conv3d = nn.Sequential(*[BasicBlock(vision_width, vision_width if i < 3 else dm, stride=1, stride_t=2,
                                                     downsample=True) for i in range(4)])
            self.conv_mu2 = nn.Conv2d(vision_width, 64, 3, 1, 1)
            self.conv_var2 = nn.Conv2d(vision_width, 64, 3, 1, 1)
            self.conv_d2 = nn.Conv2d(64, vision_width, kernel_size=3, stride=1, padding=1, bias=False)
            self.adain = ADAIN2D(vision_width, vision_width)
            if auto_beta:
                self.PID = PIDControl()
                self.KL_loss = v_kl
            else:
                self.alpha, self.beta = alpha, beta
        self.initialize_parameters()
        self.precision = "fp32"
        self.streams = 1               # >1: clip groups on concurrent HIP streams (see autoregressive_generate)
        self.ar_mode = "full"          # 'full' = the reference's per-iteration full recompute (mage_model.py:673-684);
                                       # 'incremental' = temporal KV cache, each position once
        self.last_call_mode = "eager"
        self._pad_frames: dict = {}    # zero-padded frame buffers of _frame_features (bf16 mode), by (images, device, stream)
        self.use_graph = None          # True: autoregressive_generate replays a captured HIP graph of the whole call (see _generate_graphed);
                                       # None (default): only where the call is launch-bound -- a few clips per call (_graph_auto); False: never
        self.frame_table = True        # conv3x3(token embedding) (+ in_linear) as a table sum (_frame_tables); False: the convolution GEMM
        self._graphs: dict = {}
        self._derived = _Derived(self)
        self.last_tokens: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ construction helpers
    def instantiate_first_stage(self, config):
        """Frozen, eval, train() disabled (mage_model.py:516-521)."""
        model = instantiate_from_config(config)
        self.first_stage_model = model.eval()
        self.first_stage_model.train = disabled_train.__get__(self.first_stage_model)
        for p in self.first_stage_model.parameters():
            p.requires_grad = False

    def initialize_parameters(self):
        nn.init.normal_(self.visual_token_embedding.weight, std=0.02)            # mage_model.py:524

    def set_precision(self, precision: str) -> "MAGE":
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self.precision = precision
        self.generate_model.compute_dtype, self.generate_model.split_kind = PRECISIONS[precision]
        # the once-per-clip prologue: exact-fp32 MFMA chains only in 'fp32' mode, f16x3 split operands otherwise (_lin_fp32)
        self.ma_encoder.split_kind = self.text_encoder.split_kind = 0 if precision == "fp32" else ops.F16X3
        if hasattr(self.first_stage_model, "set_precision"):     # an external latent first stage (MAGE+) has no such switch
            self.first_stage_model.set_precision(precision)
        return self

    def _dt(self) -> torch.dtype:
        return PRECISIONS[self.precision][0]

    def _sk(self) -> int:
        """Split kind of the fast parity modes (0 otherwise, and for a decoder width the split GEMM does not take)."""
        gm = self.generate_model
        return gm.split_kind if getattr(gm, "_split_on", None) and gm._split_on() else 0

    def _build(self):
        d: Dict[str, torch.Tensor] = {}
        if self.use_cids:
            d["emb"] = self.visual_token_embedding.weight.float().contiguous()
        else:
            d["emb_lin.w"] = self.visual_token_embedding.weight.float().contiguous()       # [C, embed_dim]
            d["emb_lin.b"] = self.visual_token_embedding.bias.float().contiguous()
        cw = _conv_w(self.conv[0])
        d["conv.f32"], d["conv.bf16"] = cw, cw.to(BF16)
        R, Cc = self.image_resolution, self.vision_width
        d["hwpos"] = (self.H_positional_embedding.float() + self.W_positional_embedding.float()).reshape(R * R, Cc).contiguous()
        d["speed"] = self.speed_embedding.float().reshape(-1).contiguous()
        if self.randomness:
            d["conv_d2"] = _conv_w(self.conv_d2)
            # MAGE.forward only: the Conv3d video prior, one [Cout, 3, 3, Cin] slice per temporal tap, and the two 3x3 heads
            for i, blk in enumerate(self.conv3d):
                for nm, conv in (("c1", blk.conv1), ("c2", blk.conv2), ("ds", blk.downsample[0])):
                    w = conv.weight.float()                                                # [Cout, Cin, kd, kh, kw]
                    for kd in range(3):
                        d[f"p{i}.{nm}.{kd}"] = w[:, :, kd].permute(0, 2, 3, 1).contiguous()
                for nm, gn in (("g1", blk.bn1), ("g2", blk.bn2), ("gd", blk.downsample[1])):
                    d[f"p{i}.{nm}.w"], d[f"p{i}.{nm}.b"] = gn.weight.float().contiguous(), gn.bias.float().contiguous()
            for nm, conv in (("mu2", self.conv_mu2), ("var2", self.conv_var2)):
                d[nm + ".w"], d[nm + ".b"] = _conv_w(conv), conv.bias.float().contiguous()
        return d

    # ------------------------------------------------------------------ first stage wrappers (mage_model.py:530-567)
    @torch.no_grad()
    def first_stage_encode(self, x):
        """[B, T, C, H, W] -> token ids int64 [B, T, h, w]."""
        out = self.get_first_stage_encoding(self.first_stage_model.encode(x.reshape(-1, *x.shape[-3:])))
        return out.view(*x.shape[:-3], *out.shape[1:]).contiguous()

    def get_first_stage_encoding(self, encoder_posterior):
        if isinstance(encoder_posterior, torch.Tensor):
            return encoder_posterior
        if hasattr(encoder_posterior, "sample"):          # ldm DiagonalGaussianDistribution (mage_model.py:543-544)
            return encoder_posterior.sample()
        raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")

    @torch.no_grad()
    def first_stage_decode(self, x):
        """[B, T, h, w] ids -> [B, T, C, H, W]."""
        dec = getattr(self.first_stage_model, "_decode_nocheck", self.first_stage_model.decode)   # one error check per public call
        out = dec(x.reshape(-1, *x.shape[-2:]) if self.use_cids else x.reshape(-1, *x.shape[-3:]))
        return out.view(*x.shape[:2], *out.shape[1:]).contiguous()

    # ------------------------------------------------------------------ shared pieces
    def _frame_tables(self):
        """conv3x3 over token EMBEDDINGS has only codebook_size distinct input vectors (mage_model.py:581,586-588): by linearity
            conv(emb(ids))[p] = sum_taps T[tap][ids[p + tap]],   T[tap][code] = W_tap emb[code]        [9, K, C] fp32
        and the decoder's in_linear applied to it (+ H/W positions, + bias) folds in as well:
            T2[tap][code] = W_in T[tap][code],   P2[p] = W_in (H_pos + W_pos)[p] + b_in               (mage_model.py:375-376)
        Built once per weights on the fp32 MFMA kernels (a derived cache like the folded BatchNorm vectors); 9.4 MB each at the MNIST
        config: resident in L2 / Infinity Cache.  The per-call work becomes a gather-sum (mage_table_conv): 0 matrix-core FLOPs for
        the 2*9*C^2 + 2*C^2 per pixel the reference spends (11 % of a decoder iteration), fp32-exact sums in every precision mode.
        Returns the cache dict with 'ft.T', 'ft.T2', 'ft.P2' (None when not applicable: use_cids=False, config frame_table = False); `self.frame_table = False`
        switches the callers back to the convolution GEMM + in_linear (the reference's operation order) at any time."""
        d = self._derived.get(self._build)
        if not config.get().frame_table:
            # switched off for this call only: the decision is NOT cached (a `config.override(frame_table=False)` block must not leave the tables
            # off behind it -- ADVICE r5), tables built earlier stay in the cache for when the switch is back on
            return {**d, "ft.T": None, "ft.T2": None, "ft.P2": None}
        if "ft.T" in d:
            return d
        R, Cc, Kc = self.image_resolution, self.vision_width, self.codebook_size
        gm = self.generate_model
        ok = (self.use_cids and Cc % 4 == 0
              and 9 * Kc * max(Cc, getattr(gm, "model_channels", Cc)) * 4 <= (256 << 20) and "emb" in d and d["emb"].is_cuda
              and getattr(gm, "in_channels", None) == Cc and hasattr(gm, "_build"))
        if not ok:                                  # structural: cached with the weights
            d["ft.T"] = d["ft.T2"] = d["ft.P2"] = None
            return d
        gd = gm._derived.get(gm._build)
        Cd = gm.model_channels
        emb, dev = d["emb"], d["emb"].device
        cw = d["conv.f32"].view(Cc, 9, Cc)
        T = torch.empty(9, Kc, Cc, device=dev, dtype=F32)
        T2 = torch.empty(9, Kc, Cd, device=dev, dtype=F32)
        w_in = gd["in_linear.f32"]
        for tap in range(9):
            ops.gemm(emb, cw[:, tap].contiguous(), T[tap], M=Kc, N=Cc, K=Cc, lda=Cc, ldy=Cc)
            ops.gemm(T[tap], w_in, T2[tap], M=Kc, N=Cd, K=Cc, lda=Cc, ldy=Cd)
        P2 = ops.gemm(d["hwpos"], w_in, torch.empty(R * R, Cd, device=dev, dtype=F32), M=R * R, N=Cd, K=Cc, lda=Cc, ldy=Cd, bias=gd.get("in_linear.b"))
        d["ft.T"], d["ft.T2"], d["ft.P2"] = T, T2, P2
        return d

    def _frame_source(self, tokens: torch.Tensor, dt: torch.dtype):
        """What the decoder gets for its frame slots: FrameTokens (table sum, in_linear folded) or the convolved features."""
        ft = self._frame_tables()
        if ft["ft.T2"] is not None and getattr(self, "frame_table", True):
            T2 = ft["ft.T2"]
            if dt in HALF_TYPES and self._sk() == 0:
                # bf16 / f16 mode: the table itself in that type (half the L2 / Infinity-Cache bytes per gathered row; entries rounded once,
                # sums and positions stay fp32 -- the same class of error as the 16-bit operands of the convolution it replaces)
                if "ft.T2" + _sfx(dt) not in ft:
                    ft["ft.T2" + _sfx(dt)] = T2.to(dt)
                T2 = ft["ft.T2" + _sfx(dt)]
            return FrameTokens(tokens.reshape(-1, self.image_resolution ** 2), T2, ft["ft.P2"], self.image_resolution)
        return self._frame_features(tokens, dt, split=True)

    def _frame_features(self, tokens: torch.Tensor, dt: torch.dtype, split: bool = False) -> torch.Tensor:
        """ids [n, hw] -> conv3x3(embedding) + (H_pos + W_pos) as rows [n*hw, C] (mage_model.py:581,586-588,674-676).

        bf16 mode: the embedding rows are written into the interior of a zero-padded (R+2) x (R+2) frame buffer, so that the
        convolution is the padded-taps form of mage_gemm (every tap a valid row: the 8-phase ping-pong kernel with one scalar
        offset per K slab, the positional table loaded into the accumulators) instead of the generic per-lane gather.  The buffer
        is kept between calls: its border is written once (zeros), its interior on every call."""
        d = self._derived.get(self._build)
        R, Cc = self.image_resolution, self.vision_width
        n = tokens.numel() // (R * R)
        if dt == F32 and not split:
            ft = self._frame_tables()
            if ft["ft.T"] is not None and getattr(self, "frame_table", True):                  # the once-per-clip prologue: fp32 features
                return ops.table_conv(tokens.reshape(-1).contiguous(), ft["ft.T"], torch.empty(n * R * R, Cc, device=tokens.device, dtype=F32),
                                      n_img=n, H=R, W=R, pos=d["hwpos"])
        sk = self._sk() if split else 0
        if sk and Cc % 256 == 0 and (n * R * R) % 256 == 0 and self.generate_model._taps_ok(n * R * R):
            # fast parity modes: the same padded-taps convolution on split-precision operands; the features leave as split rows
            # (in_linear's A operand)
            P = R + 2
            key = (n, str(tokens.device), torch.cuda.current_stream(tokens.device).cuda_stream, sk)
            pad = self._pad_frames.get(key)
            if pad is None:
                if len(self._pad_frames) > 8:
                    self._pad_frames.clear()
                pad = self._pad_frames[key] = ops.split_empty(n * P * P + 1, Cc, sk, tokens.device, zero=True)
            ops.embedding(tokens.reshape(-1), d["emb"], pad, group=R * R, group_stride=P * P, off=P + 1, inner=R, inner_stride=P, split_kind=sk)
            out = ops.split_empty(n * R * R, Cc, sk, tokens.device)
            return ops.gemm(pad, _wsplit(d, "conv", sk), out, M=n * R * R, N=Cc, K=9 * Cc, lda=2 * Cc, ldy=2 * Cc, out_h=R, out_w=R, in_h=P,
                            in_w=P, a_img_stride=P * P, taps_h=3, taps_w=3, cin=Cc, rowadd=d["hwpos"], rowadd_div=1, rowadd_mod=R * R,
                            split_kind=sk, y_split=True)
        if dt == F32 or Cc % 64:
            emb = ops.embedding(tokens.reshape(-1), d["emb"], torch.empty(n * R * R, Cc, device=tokens.device, dtype=dt))
            return VectorQuantizedVAE._conv(emb, _wdt(d, "conv", dt), torch.empty_like(emb), n_img=n, H=R, W=R, cin=Cc, cout=Cc,
                                            k=3, rowadd=d["hwpos"], rowadd_div=1, rowadd_mod=R * R)
        P = R + 2
        key = (n, str(tokens.device), torch.cuda.current_stream(tokens.device).cuda_stream if tokens.is_cuda else 0)
        pad = self._pad_frames.get(key)
        if pad is None:
            if len(self._pad_frames) > 8:
                self._pad_frames.clear()
            pad = self._pad_frames[key] = torch.zeros((n * P * P + 1) * Cc, device=tokens.device, dtype=dt).view(-1, Cc)
        ops.embedding(tokens.reshape(-1), d["emb"], pad, group=R * R, group_stride=P * P, off=P + 1, inner=R, inner_stride=P)
        out = torch.empty(n * R * R, Cc, device=tokens.device, dtype=dt)
        return ops.gemm(pad, _wdt(d, "conv", dt), out, M=n * R * R, N=Cc, K=9 * Cc, lda=Cc, ldy=Cc, out_h=R, out_w=R, in_h=P, in_w=P,
                        a_img_stride=P * P, taps_h=3, taps_w=3, cin=Cc, stride=1, dy0=0, dx0=0, rowadd=d["hwpos"], rowadd_div=1,
                        rowadd_mod=R * R)

    def _frame_features_latent(self, lat: torch.Tensor, ld: int, dt: torch.dtype) -> torch.Tensor:
        """use_cids=False: latents as fp32 rows [n*hw, ld] (first embed_dim columns valid) -> Linear(embed_dim -> C)
        (mage_model.py:483,583,646) -> conv3x3 + (H_pos + W_pos), rows [n*hw, C]."""
        d = self._derived.get(self._build)
        R, Cc = self.image_resolution, self.vision_width
        rows = lat.numel() // ld
        E = d["emb_lin.w"].shape[1]
        # K = 4: fp32 MFMA path; it writes fp32 or bf16 rows -- the f16 mode takes fp32 rows through one cast pass
        emb = ops.gemm(lat, d["emb_lin.w"], torch.empty(rows, Cc, device=lat.device, dtype=F32 if dt == F16 else dt), M=rows, N=Cc, K=E, lda=ld,
                       ldy=Cc, bias=d["emb_lin.b"])
        if dt == F16:
            emb = ops.cast(emb, torch.empty(rows, Cc, device=lat.device, dtype=F16))
        return VectorQuantizedVAE._conv(emb, _wdt(d, "conv", dt), torch.empty_like(emb), n_img=rows // (R * R), H=R, W=R, cin=Cc,
                                        cout=Cc, k=3, rowadd=d["hwpos"], rowadd_div=1, rowadd_mod=R * R)

    def _motion_anchor(self, tok0: torch.Tensor, batch, noise: Optional[torch.Tensor], first: Optional[torch.Tensor] = None,
                       video_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Once-per-clip prologue, fp32 (mage_model.py:648-668 / 586-613): rows [B*hw, C].  video_rows [B*hw, 64] (forward with
        randomness: the reparameterised video prior, before conv_d2) replaces the sampled noise."""
        d = self._derived.get(self._build)
        B = batch["text"].shape[0]
        R, Cc = self.image_resolution, self.vision_width
        if first is None:
            first = self._frame_features(tok0, F32)                                           # [B*hw, C]
        txt = self.text_encoder(batch["text"])                                                # [B, S, C]
        S = txt.shape[1]
        ma = self.ma_encoder._run(first, txt.reshape(B * S, -1), B=B, nq=R * R, nk=S, seq_first=False)
        if self.randomness:
            if video_rows is not None:                                                        # forward(): reparameterised prior
                nz = video_rows
            else:
                if noise is None:
                    noise = torch.randn([B, 64, R, R], device=ma.device)                      # mage_model.py:661
                nz = noise.float().permute(0, 2, 3, 1).reshape(B * R * R, 64).contiguous()
            y = VectorQuantizedVAE._conv(nz, d["conv_d2"], torch.empty(B * R * R, Cc, device=ma.device, dtype=F32), n_img=B,
                                         H=R, W=R, cin=64, cout=Cc, k=3)
            ma = self.adain._run(ma, y, B, R, R)
        if "speed" in batch:
            ops.add_scaled_rowvec(ma, batch["speed"].float().contiguous(), d["speed"], B=B, P=R * R, Cc=Cc)
        return ma

    # ------------------------------------------------------------------ sampling (mage_model.py:641-693)
    @torch.no_grad()
    def autoregressive_generate(self, batch):
        """batch {'images' [B,L,C,H,W] (only frame 0 is read), 'text' int64 [B,S], 'speed' [B] optional,
        'video_noise' [B,64,h,w] optional (injects the randomness-branch noise instead of torch.randn)} -> [B,L,C,H,W].

        With ``self.streams = n > 1`` the clips are processed as n independent groups on n HIP streams: clips never
        interact, so the results are bit-identical, and the HBM-bound kernels of one group (LayerNorm, attention, casts)
        run under the MFMA-bound GEMMs of the other instead of in front of them."""
        images = batch["images"]
        _need_gpu(images, "MAGE.autoregressive_generate")
        # (weights_frozen: no parameter changes during one inference call -- the derived caches validate once, not at each of their ~40 fetches)
        with torch.cuda.device(images.device), weights_frozen():
            ug = self._graph_auto(batch) if self.use_graph is None else bool(self.use_graph)
            # (clip groups on several streams are captured too when `graph_multistream` is set: fork / join edges inside ONE graph -- the probe of
            # tools/inc_streams_graph_probe.py; off by default: measured slower than one stream at every size, DESIGN finding 94)
            if (ug and (int(getattr(self, "streams", 1)) == 1 or getattr(self, "graph_multistream", False))
                    and not torch.cuda.is_current_stream_capturing()):
                out = self._generate_graphed(batch)
            else:
                self.last_call_mode = "eager"
                out = self._generate_eager(batch)
            ops.check_device_errors(images.device)        # e.g. a caption id >= vocab_size: the reference raises IndexError
        return out

    def _graph_auto(self, batch) -> bool:
        """use_graph = None: replay from a captured graph where the call is launch-bound -- up to 4 clips per call (the reference samples
        ONE, main_mage.py:205: ~870 launches of 3-15 µs; 13.7 ms eager vs 7.6 ms replayed per 16-frame clip), the VQ-token path only (an
        external latent first stage is not ours to capture), and not while per-launch profiling is on."""
        images = batch["images"]
        return (self.use_cids and images.shape[0] * self.image_resolution ** 2 <= 1024 and not ops.PROFILE.enabled
                and all(torch.is_tensor(v) for v in batch.values()) and config.get().auto_graph)

    def _graph_fingerprint(self):
        """Everything besides shapes, precision, AR mode and weights that selects kernels: a captured graph replays only under the same."""
        fs = self.first_stage_model
        return (int(getattr(self, "streams", 1)), bool(getattr(self, "frame_table", True)), self.generate_model._stream_bf16(), bool(getattr(self.ma_encoder, "mage_plus", False)),
                getattr(self.ma_encoder, "split_kind", 0), getattr(self.text_encoder, "split_kind", 0),
                tuple(str(getattr(fs, a, None)) for a in ("decode_dtype", "encode_split", "decode_split")),
                config.get(), tuple(sorted(config.lib_options().items())))

    def _generate_eager(self, batch):
        if not self.use_cids:
            return self._generate_latent(batch)
        n = int(getattr(self, "streams", 1))
        if n > 1 and batch["images"].shape[0] >= 2 * n and batch["images"].shape[0] % n == 0:
            return self._generate_multistream(batch, n)
        return self._generate_one(batch)

    def _generate_graphed(self, batch):
        """HIP-graph replay of the whole call ("whole-loop residency").  The call is a fixed sequence of
        kernel launches on device-resident buffers -- no host decision depends on device data (the argmax of frame i is written
        into slot i+1 of the token buffer on the device) -- so one capture of the launch stream replays it with ONE host call
        instead of ~50 per AR iteration: the Python / ctypes launch loop (14 of 30 ms per call in incremental mode) disappears.
        First call with a given (shapes, precision, AR mode, weights): eager, which also builds every derived cache; second:
        captured on static copies of the inputs (all intermediates live in the graph's private pool: an arena that is never
        re-allocated); from then on: inputs are copied into the static buffers, the graph is replayed, the video and tokens are
        cloned out (last_logits too).  Same kernels, same order, same
        arithmetic: bit-identical to the eager call."""
        self._warm_derived()
        gens = tuple(getattr(m, "_derived").gen for m in (self, self.generate_model, self.ma_encoder, self.text_encoder,
                                                          getattr(self, "adain", None), self.first_stage_model)
                     if m is not None and hasattr(m, "_derived"))
        prof = ops.PROFILE.mode_key()
        self.last_call_mode = "eager"
        if prof != "off" and not ops.graph_events_supported(batch["images"].device):
            return self._generate_eager(batch)              # per-launch events wanted, but they cannot be captured here
        key = (tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items())), self.precision, self.ar_mode, gens, prof,
               str(batch["images"].device), self._graph_fingerprint())
        ent = self._graphs.get(key)
        if ent is None:
            self._graphs = {k: v for k, v in self._graphs.items() if k[3] == gens}       # graphs of replaced weights are dead
            while len(self._graphs) >= 4:                                                # each graph owns an arena: keep the 4 newest shapes
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = "warm" if self.use_graph else "warm2"
            return self._generate_eager(batch)
        if ent == "off":
            return self._generate_eager(batch)
        if ent == "warm2":                               # auto mode: a shape has to come back twice before it is worth a capture
            self._graphs[key] = "warm"
            return self._generate_eager(batch)
        if ent == "warm":
            static = {k: v.clone() for k, v in batch.items()}
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            saved = ops.PROFILE.capture_begin()
            # no cyclic garbage collection while the stream is capturing: a collection that frees device memory or another captured
            # graph in the middle of a capture (older models of the same process going away) aborts the process
            gc_was = gc.isenabled()
            gc.collect()
            gc.disable()
            try:
                with torch.cuda.graph(g):
                    out = self._generate_eager(static)
                    toks, logits = self.last_tokens, self.last_logits
            except Exception:
                if self.use_graph:                       # asked for explicitly: loud
                    raise
                self._graphs[key] = "off"                # auto mode: this call shape cannot be captured here -- eager from now on
                torch.cuda.synchronize()
                return self._generate_eager(batch)
            finally:
                if gc_was:
                    gc.enable()
                recs = ops.PROFILE.capture_end(saved)
            ent = self._graphs[key] = {"g": g, "in": static, "out": out, "tok": toks, "logits": logits, "recs": recs}
        else:
            for k, v in batch.items():
                ent["in"][k].copy_(v)
        ent["g"].replay()
        self.last_call_mode = "graph"
        if ent["recs"] and ops.PROFILE.enabled:
            torch.cuda.current_stream().synchronize()
            ops.PROFILE.absorb(ent["recs"])
        self.last_tokens = None if ent["tok"] is None else ent["tok"].clone()
        # cloned like the tokens and the output: graph replay is the DEFAULT for small calls (use_graph = None), and a caller that keeps
        # last_logits across two generations must not find the first one overwritten by the second (0.5 MB per clip at cfg2)
        self.last_logits = None if ent["logits"] is None else ent["logits"].clone()
        return ent["out"].clone()

    def _warm_derived(self) -> None:
        """Build every derived weight cache (bf16 copies, transposed codebook, folded BatchNorm vectors, summed positional
        tables) on the CALLER's stream: a cache built lazily inside one side stream would be read by the others with no event
        in between."""
        self._derived.get(self._build)
        for mod in (self.generate_model, self.ma_encoder, self.text_encoder, getattr(self, "adain", None), self.first_stage_model):
            if mod is not None and hasattr(mod, "_derived") and hasattr(mod, "_build"):
                mod._derived.get(mod._build)
        self._frame_tables()
        for enc, names in ((self.ma_encoder, [f"b{i}.{n}" for i in range(self.ma_encoder.layers) for n in ("in_proj", "out_proj", "c_fc", "c_proj")]),
                           (self.text_encoder, [f"l{i}.{n}" for i in range(self.text_encoder.transformer_layers)
                                                for n in ("in_proj", "out_proj", "fc2")] + ["proj"])):
            if getattr(enc, "split_kind", 0):
                de = enc._derived.get(enc._build)
                for n in names:
                    if de[n + ".f32"].shape[1] % 64 == 0:
                        _wsplit(de, n, enc.split_kind)
        if hasattr(self.generate_model, "_warm_split"):
            self.generate_model._warm_split()
            if self._sk():
                _wsplit(self._derived.get(self._build), "conv", self._sk())

    def _generate_multistream(self, batch, n):
        B = batch["images"].shape[0]
        per = B // n
        self._warm_derived()                                          # before the fork: side streams only READ the caches
        main = torch.cuda.current_stream(batch["images"].device)
        if getattr(self, "_side_streams", None) is None or len(self._side_streams) != n:
            self._side_streams = [torch.cuda.Stream(device=batch["images"].device) for _ in range(n)]
        outs, toks = [None] * n, [None] * n
        for g, st in enumerate(self._side_streams):
            st.wait_stream(main)                                      # inputs were produced on the caller's stream
            with torch.cuda.stream(st):
                sub = {k: v[g * per:(g + 1) * per] for k, v in batch.items()}
                outs[g] = self._generate_one(sub)
                toks[g] = self.last_tokens
        for st in self._side_streams:
            main.wait_stream(st)
        for t_ in outs + toks:
            t_.record_stream(main)                                    # allocator plumbing: consumed on the caller's stream
        self.last_tokens, self.last_logits = torch.cat(toks, 0), None
        return torch.cat(outs, 0)

    @torch.no_grad()
    def _generate_latent(self, batch):
        """use_cids=False (MAGE+, mage_model.py:645-646,683-684,689): the first stage is an external latent autoencoder
        (ldm AutoencoderKL in config/mage+_*.yaml); its encode/decode are the boundary, everything between runs here.
        The GroupNorm head mixes all frames of a clip, so only the reference's full-recompute loop is valid."""
        images = batch["images"]
        B = images.shape[0]
        R, L = self.image_resolution, self.frames_length
        hw, Lm1 = R * R, L - 1
        dt = self._dt()
        E = self.first_stage_model.embed_dim
        lat0 = self.first_stage_encode(images[:, 0:1])[:, 0]                                   # [B, E, h, w]
        LD = 8                                                                                # row stride of the latent buffer
        cur = torch.zeros(B, Lm1, hw, LD, device=images.device, dtype=F32)
        cur[..., :E] = lat0.permute(0, 2, 3, 1).reshape(B, 1, hw, E).float()                  # :670 every slot holds frame 0
        first = self._frame_features_latent(cur[:, 0].contiguous(), LD, F32)
        ma = self._motion_anchor(None, batch, batch.get("video_noise"), first=first)
        ma_dt = _to_dt(ma, dt)
        pred = None
        for i in range(Lm1):                                                                  # :673-684
            feats = self._frame_features_latent(cur.view(-1, LD), LD, dt)
            pred = self.generate_model._run(ma_dt, feats, B=B, hh=R, ww=R).view(B, Lm1, hw, -1)
            if i != Lm1 - 1:
                cur[:, i + 1, :, :E] = pred[:, i, :, :E]                                      # :684 (index plumbing)
        self.last_tokens, self.last_logits = None, pred[..., :E].reshape(B, Lm1, R, R, E)
        gen = pred[..., :E].reshape(B * Lm1, R, R, E).permute(0, 3, 1, 2).contiguous()          # :689
        video = self.first_stage_model.decode(gen)
        video = video.view(B, Lm1, *video.shape[1:])
        return _assemble(images, video)

    @torch.no_grad()
    def _generate_one(self, batch):
        images = batch["images"]
        B = images.shape[0]
        R, L, K = self.image_resolution, self.frames_length, self.codebook_size
        hw, Lm1 = R * R, self.frames_length - 1
        dt = self._dt()
        tok0 = self.first_stage_encode(images[:, 0:1])[:, 0].reshape(B, hw)                   # :642
        ma = self._motion_anchor(tok0, batch, batch.get("video_noise"))
        ma_dt = _to_dt(ma, dt)
        gen = torch.empty(B, Lm1, R, R, device=images.device, dtype=torch.int64)
        if self.ar_mode == "incremental":
            # each position once, temporal K,V cached; bit-identical tokens to the reference loop below
            st = self.generate_model._inc_begin(B, R, R, images.device)
            prev = tok0.contiguous()
            gen_t = torch.empty(Lm1, B, hw, device=images.device, dtype=torch.int64)         # frame-major: a frame's tokens are contiguous,
            for i in range(Lm1):                                                              # the argmax writes them where the next step reads them
                feats = self._frame_source(prev, dt)                                          # newest frame only
                step_logits = self.generate_model._inc_step(st, ma_dt if i == 0 else None, feats)
                prev = gen_t[i]
                ops.argmax(step_logits, prev, rows=B * hw, K=K)
            gen = gen_t.permute(1, 0, 2).reshape(B, Lm1, R, R) if B == 1 else gen_t.permute(1, 0, 2).contiguous().view(B, Lm1, R, R)   # index plumbing, once
            self.last_tokens, self.last_logits = gen, None
            video = self.first_stage_decode(gen)
            return _assemble(images, video)
        cur = tok0[:, None, :].repeat(1, Lm1, 1).contiguous()                                 # :670 future slots hold frame 0
        logits = None
        for i in range(Lm1):                                                                  # :673-684
            feats = self._frame_source(cur, dt)
            ev = ops.PROFILE.begin() if ops.PROFILE.wants("decoder_step") else None          # bench.py: the transformer step on its own
            logits = self.generate_model._run(ma_dt, feats, B=B, hh=R, ww=R)                  # [B*(L-1)*hw, K]
            if ev is not None:
                ops.PROFILE.end("decoder_step", ev, 0.0)
            if i != Lm1 - 1:                                                                  # argmax of frame i -> slot i+1
                ops.argmax(logits, cur, rows=B * hw, K=K, group=hw, in_group_stride=Lm1 * hw, in_off=i * hw,
                           out_group_stride=Lm1 * hw, out_off=(i + 1) * hw)
        ops.argmax(logits, gen, rows=B * Lm1 * hw, K=K)                                       # :687
        self.last_tokens, self.last_logits = gen, logits.view(B, Lm1, R, R, K)
        video = self.first_stage_decode(gen)                                                  # :690
        return _assemble(images, video)                         # :691

    # ------------------------------------------------------------------ teacher-forced pass (mage_model.py:575-639)
    @torch.no_grad()
    def _video_prior(self, tok: Optional[torch.Tensor], lat_rows: Optional[torch.Tensor] = None, B: int = 0, L: int = 0,
                     tape: Optional[list] = None, dt: torch.dtype = F32) -> torch.Tensor:
        """self.conv3d over the token embeddings of ALL frames (mage_model.py:496-501,602-603): tok int64 [B, L, hw] (or, for
        use_cids=False, lat_rows [B*L*hw, 8] fp32 latents whose Linear(E -> C) embedding is taken) -> rows [B*hw, d_model] fp32.  Each Conv3d(3x3x3, temporal stride s, pad 1) is three implicit-GEMM launches, one per temporal tap,
        accumulating in place: the block input lives in a zero-padded frame buffer in which clip b owns frames
        [b*Lp, (b+1)*Lp) (frame 0 = the leading zero pad) with Lp = s * (virtual output frames per clip), so that output image
        i' = b*Lv + t' gathers from frame s*i' + kd: one affine image stride for the whole batch, the clip boundaries and the
        temporal padding are zero frames.  The Lv - Lout virtual frames per clip are never normalised or read.
        `tape` (a list, training path): receives per block what the backward pass needs (modules/mage_train_prior.py).
        dt: storage / MFMA dtype of the activations and weights (bf16 in bf16 training: the convolution outputs, the GroupNorm
        statistics and the block's result stay fp32; the values path and fp32 mode run everything in fp32)."""
        d = self._derived.get(self._build)
        R, Cc = self.image_resolution, self.vision_width
        hw = R * R
        if tok is not None:
            B, L = tok.shape[0], tok.shape[1]
        dev = tok.device if tok is not None else lat_rows.device

        def wdt(key):
            if dt == F32:
                return d[key]
            if key + ".bf16" not in d:
                d[key + ".bf16"] = d[key].to(BF16)
            return d[key + ".bf16"]

        def conv3(xpad, key, Lv, s_t, cin, cout):
            out = torch.empty(B * Lv * hw, cout, device=dev, dtype=F32)
            for kd in range(3):
                ops.gemm(xpad, wdt(f"{key}.{kd}"), out, M=B * Lv * hw, N=cout, K=9 * cin, lda=cin, ldy=cout, out_h=R, out_w=R,
                         in_h=R, in_w=R, taps_h=3, taps_w=3, cin=cin, stride=1, dy0=-1, dx0=-1, a_img_stride=s_t * hw,
                         a_off=kd * hw, residual=out if kd else None, ldr=cout)
            return out

        Lin = L
        Lout = (Lin + 1) // 2
        Lp = 2 * (Lout + 1)
        xa = torch.zeros((B * Lp + 1) * hw, Cc, device=dev, dtype=dt)                         # block 0 input: the embeddings
        if tok is not None:
            ops.embedding(tok.reshape(-1).contiguous(), d["emb"], xa, group=L * hw, group_stride=Lp * hw, off=hw)
        else:                                                      # Linear(E -> C) straight into the padded frame buffer (:583)
            E = d["emb_lin.w"].shape[1]
            ops.gemm(lat_rows, d["emb_lin.w"], xa, M=B * L * hw, N=Cc, K=E, lda=lat_rows.shape[1], ldy=Cc, out_w=L * hw,
                     y_img_stride=Lp * hw, y_off=hw, bias=d["emb_lin.b"])
        cin = Cc
        for i, blk in enumerate(self.conv3d):
            cout = blk.conv1.out_channels
            Lout = (Lin + 1) // 2
            Lv = Lout + 1
            gn = dict(n_samples=B, rows_per_sample=Lout * hw, groups=16)
            st1, std, st2 = (torch.empty(B, 16, 2, device=dev, dtype=F32) for _ in range(3))
            c1 = conv3(xa, f"p{i}.c1", Lv, 2, cin, cout)
            cd = conv3(xa, f"p{i}.ds", Lv, 2, cin, cout)
            xb = torch.zeros((B * (Lout + 2) + 2) * hw, cout, device=dev, dtype=dt)            # conv2 input: stride-1 layout
            ops.groupnorm_act(c1, d[f"p{i}.g1.w"], d[f"p{i}.g1.b"], xb, sample_stride_rows=Lv * hw, row_off=0, eps=blk.bn1.eps,
                              act=1, y_sample_stride_rows=(Lout + 2) * hw, y_row_off=hw, stats=st1, **gn)
            res = ops.groupnorm_act(cd, d[f"p{i}.gd.w"], d[f"p{i}.gd.b"], torch.empty(B * Lout * hw, cout, device=dev, dtype=F32),
                                    sample_stride_rows=Lv * hw, row_off=0, eps=blk.downsample[1].eps, act=0, stats=std, **gn)
            c2 = conv3(xb, f"p{i}.c2", Lout + 2, 1, cout, cout)
            if i + 1 < len(self.conv3d):                                                       # next block's stride-2 input
                Lp2 = 2 * ((Lout + 1) // 2 + 1)
                nxt = torch.zeros((B * Lp2 + 1) * hw, cout, device=dev, dtype=dt)
                ops.groupnorm_act(c2, d[f"p{i}.g2.w"], d[f"p{i}.g2.b"], nxt, sample_stride_rows=(Lout + 2) * hw, row_off=0,
                                  eps=blk.bn2.eps, act=1, residual=res, y_sample_stride_rows=Lp2 * hw, y_row_off=hw, stats=st2, **gn)
                out_map = (Lp2 * hw, hw)
            else:
                if Lout != 1:
                    raise ValueError(f"the Conv3d video prior's four stride-2 blocks must collapse the clip to ONE frame "
                                     f"(frames_length <= 16); got {L} frames -> {Lout}")
                nxt = ops.groupnorm_act(c2, d[f"p{i}.g2.w"], d[f"p{i}.g2.b"], torch.empty(B * hw, cout, device=dev, dtype=F32),
                                        sample_stride_rows=(Lout + 2) * hw, row_off=0, eps=blk.bn2.eps, act=1, residual=res, stats=st2,
                                        **gn)
                out_map = (hw, 0)
            if tape is not None:
                tape.append(dict(xa=xa, c1=c1, cd=cd, xb=xb, c2=c2, res=res, st1=st1, std=std, st2=st2, Lin=Lin, Lout=Lout, Lv=Lv, cin=cin,
                                 cout=cout, out_map=out_map, dt=dt))
            xa, cin, Lin = nxt, cout, Lout
        return xa

    def teacher_forced_logits(self, batch, extras: Optional[dict] = None):
        """tokens [B, L, h, w] and logits [B, L-1, h, w, K] of one teacher-forced decoder pass.  With randomness=True the motion
        anchor is modulated by the reparameterised Conv3d video prior (mage_model.py:601-609); `extras` (a dict) then receives
        'kl_sum' [B] (sum of 1 + logvar - mu^2 - exp(logvar) per sample) and 'prior' rows [B*hw, C]."""
        images = batch["images"]
        _need_gpu(images, "MAGE.forward")
        if not self.use_cids:
            raise NotImplementedError("teacher_forced_logits is the use_cids=True path; MAGE.forward handles use_cids=False")
        B = images.shape[0]
        R, L = self.image_resolution, self.frames_length
        dt = self._dt()
        tok = self.first_stage_encode(images).reshape(B, -1, R * R)                          # :579
        video_rows = self._reparam_video_rows(batch, B, extras, tok=tok) if self.randomness else None
        ma = self._motion_anchor(tok[:, 0].contiguous(), batch, None, video_rows=video_rows)
        feats = self._frame_source(tok[:, :L - 1].contiguous(), dt)
        logits = self.generate_model._run(_to_dt(ma, dt), feats, B=B, hh=R, ww=R)
        return tok.view(B, -1, R, R), logits.view(B, L - 1, R, R, self.codebook_size)

    def _reparam_video_rows(self, batch, B: int, extras: Optional[dict], tok=None, lat_rows=None, L: int = 0) -> torch.Tensor:
        """Conv3d video prior -> conv_mu2 / conv_var2 -> reparameterisation (mage_model.py:601-604,569-573): rows [B*hw, 64]
        (before conv_d2); extras receives 'kl_sum' [B] and 'prior' rows."""
        R = self.image_resolution
        d = self._derived.get(self._build)
        prior = self._video_prior(tok, lat_rows, B, L)                                    # :602-603
        Cp, dev = prior.shape[1], prior.device
        mu = VectorQuantizedVAE._conv(prior, d["mu2.w"], torch.empty(B * R * R, 64, device=dev, dtype=F32), n_img=B, H=R, W=R,
                                      cin=Cp, cout=64, k=3, bias=d["mu2.b"])              # :570
        logvar = VectorQuantizedVAE._conv(prior, d["var2.w"], torch.empty_like(mu), n_img=B, H=R, W=R, cin=Cp, cout=64, k=3,
                                          bias=d["var2.b"])
        eps = batch.get("reparam_noise")                                                  # [B,64,h,w]; else torch.randn (:571)
        if eps is None:
            eps = torch.randn(B, 64, R, R, device=dev)
        eps = eps.to(dev).float().permute(0, 2, 3, 1).reshape(B * R * R, 64).contiguous()
        kl_sum = torch.empty(B, device=dev, dtype=F32)
        video_rows = ops.reparam_kl(mu.view(B, -1), logvar.view(B, -1), eps.view(B, -1), torch.empty_like(mu).view(B, -1), kl_sum)
        video_rows = video_rows.view(B * R * R, 64)
        if extras is not None:
            extras.update(kl_sum=kl_sum, prior=prior)
        if batch.get("_test_flag"):
            # forward(test_flag=True) (mage_model.py:604-605): mu / logvar still feed the KL term, the embedding itself is fresh noise
            # (batch['video_noise'] [B,64,h,w] injects it, as in autoregressive_generate)
            nz = batch.get("video_noise")
            if nz is None:
                nz = torch.randn(B, 64, R, R, device=dev)
            video_rows = nz.to(dev).float().permute(0, 2, 3, 1).reshape(B * R * R, 64).contiguous()
        return video_rows

    @torch.no_grad()
    def _forward_latent(self, batch, extras: dict):
        """use_cids=False (MAGE+, mage_model.py:579,583,620): latents of ALL frames from the external first stage, teacher-forced
        latent prediction, MSE.  Returns (mse 0-dim tensor, pred rows [B*(L-1)*hw, 8])."""
        images = batch["images"]
        B = images.shape[0]
        R, L = self.image_resolution, self.frames_length
        hw, dt = R * R, self._dt()
        E, LD = self.first_stage_model.embed_dim, 8
        lat = self.first_stage_encode(images)                                                 # [B, L, E, h, w]
        Lb = lat.shape[1]
        rows = torch.zeros(B, Lb, hw, LD, device=images.device, dtype=F32)
        rows[..., :E] = lat.permute(0, 1, 3, 4, 2).reshape(B, Lb, hw, E).float()              # layout plumbing (channels last)
        video_rows = None
        if self.randomness:
            video_rows = self._reparam_video_rows(batch, B, extras, lat_rows=rows.view(-1, LD), L=Lb)
        first = self._frame_features_latent(rows[:, 0].contiguous().view(-1, LD), LD, F32)
        ma = self._motion_anchor(None, batch, None, first=first, video_rows=video_rows)
        feats = self._frame_features_latent(rows[:, :L - 1].contiguous().view(-1, LD), LD, dt)
        pred = self.generate_model._run(_to_dt(ma, dt), feats, B=B, hh=R, ww=R)      # [B*(L-1)*hw, 8]
        tgt = rows[:, 1:L].contiguous().view(-1, LD)
        return ops.mse(pred, tgt, rows=B * (L - 1) * hw, cols=E, lda=pred.shape[1], ldb=LD), pred

    def _forward_with_graph(self, batch):
        """Grad mode (the training loop, main_mage.py:150-153): the same pass as one autograd node whose inputs are the trainable
        parameters (mage_train.MageLossFn), so that ``loss.backward()`` fills every ``.grad`` from the HIP backward kernels."""
        from . import mage_train
        _need_gpu(batch["images"], "MAGE.forward")
        with torch.cuda.device(batch["images"].device):
            names = mage_train.trainable_names(self)
            byname = dict(self.named_parameters())
            loss = mage_train.MageLossFn.apply(self, batch, names, *[byname[n] for n in names])
            prefix = "train" if self.training else "val"
            ops.check_device_errors(batch["images"].device)
        return loss, {f"{prefix}/{k}": v for k, v in self._last_train_parts.items()}

    def forward(self, batch, test_flag=False):
        """(loss, loss_dict) of the teacher-forced pass (mage_model.py:575-639), incl. the randomness=True terms (KL of the
        reparameterised video prior, the PID-controlled or fixed beta, the speed-embedding l2).  batch['reparam_noise']
        [B,64,h,w] optionally injects the reparameterisation noise; ``test_flag=True`` (values only) replaces the video embedding by noise
        (batch['video_noise'] injects it) while the KL term still comes from mu / logvar (:604).  Under ``torch.no_grad()``: values only.  In grad mode (any
        parameter requiring grad): the returned loss carries an autograd node backed by the HIP backward kernels
        (modules/mage_train.py, mage_train_prior.py: every config family -- MNIST, CATER with the randomness branch, MAGE+), so
        ``loss.backward(); optimizer.step()`` works."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if test_flag and self.randomness:
                raise NotImplementedError("forward(test_flag=True) is an evaluation switch (mage_model.py:604): call it under torch.no_grad()")
            return self._forward_with_graph(batch)
        batch = dict(batch, _test_flag=bool(test_flag and self.randomness))      # :604: the video embedding is replaced by noise
        extras: dict = {}
        L = self.frames_length
        if self.use_cids:
            tok, logits = self.teacher_forced_logits(batch, extras)
            recon = ops.cross_entropy(logits.reshape(-1, self.codebook_size), tok[:, 1:L].reshape(-1).contiguous())   # :618
        else:
            _need_gpu(batch["images"], "MAGE.forward")
            recon, self.last_logits = self._forward_latent(batch, extras)                                            # :620
        prefix = "train" if self.training else "val"
        ld = {f"{prefix}/prediction": recon.item()}
        final = recon
        if self.randomness:
            kl = -0.5 * extras["kl_sum"].mean()                                               # :623 (a [B]-element reduction)
            ld[f"{prefix}/kl_loss"] = kl.item()
            if self.auto_beta:
                self.beta, _ = self.PID.pid(self.KL_loss, kl.item())                          # :627
                ld[f"{prefix}/beta"] = self.beta
                final = recon + self.beta * kl
            else:
                # mean_b || speed_b * speed_embedding ||^2 (:631); like the reference this branch needs batch['speed']
                l2 = (batch["speed"].float().to(recon.device) ** 2).mean() * (self.speed_embedding.float() ** 2).sum()
                final = recon + self.beta * kl + self.alpha * l2
        ld[f"{prefix}/final_loss"] = final.item()
        ops.check_device_errors(batch["images"].device)
        return final, ld
