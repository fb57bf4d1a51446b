"""Training path of MAGE on the HIP kernels: ``loss, _ = model(batch); loss.backward()``.

The reference trains with ``loss.backward(); optimizer.step()`` (main_mage.py:150-153) through PyTorch autograd.  Here the
whole teacher-forced pass is ONE ``torch.autograd.Function``: ``forward`` runs the same libmage_hip.so kernels as inference
while keeping what the backward pass needs (the LayerNorm inputs, qkv, attention outputs, MLP pre-activations), ``backward``
walks the graph by hand and hands autograd one gradient per parameter -- so ``.grad`` fields, optimizers and
``DistributedDataParallel`` hooks see exactly what they see with the reference.

Dense gradients are MFMA GEMMs: dX = dY W runs the forward kernel on a transposed weight copy; dW = dY^T X contracts over all
tokens into a small [N, K] matrix, so both operands are transposed once (mage_transpose) and multiplied in ONE split-K launch
(mage_gemm n_split) whose partials are summed in a fixed order.  LayerNorm / attention / activation / cross-entropy backward,
the embedding scatter and the positional-table reductions are the kernels of csrc/train.hip.

Built for every config family of the reference: the MNIST configs of BASELINE cfg1-3; with the randomness branch of
modules/mage_train_prior.py (Conv3d video prior, reparameterisation + KL, ADAIN) config/mage_caterv1.yaml / mage_caterv2.yaml; and
MAGE+ (config/mage+_*.yaml: use_cids=False -- Linear token embedding of the first stage's latents, GroupNorm + SiLU + Conv3d head,
MSE loss, the ln_q / ln_kv TransformerBlock variant, PID-controlled beta).  The first stage is frozen, as in the reference
(mage_model.py:516-521).  Dropout (training mode only) is a stateless mask recomputed in the
backward pass from a per-call seed.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import config, ops
from .vqvae_model import VectorQuantizedVAE

F32 = torch.float32
BF16 = torch.bfloat16

__all__ = ["MageLossFn", "train_forward", "train_backward", "trainable_names"]


def _sfx(dt):
    return ".f32" if dt == F32 else ".bf16"


def _wt(d: Dict[str, torch.Tensor], name: str, dt) -> torch.Tensor:
    """Transposed copy [K, N] of the Linear weight `name` ([N, K]) in dt: the W operand of dX = dY W (a derived cache like the
    bf16 copies; rebuilt with them when a parameter changes)."""
    key = name + ".T" + _sfx(dt)
    if key not in d:
        d[key] = d[name + _sfx(dt)].t().contiguous()
    return d[key]


class _Run:
    """Per-call state: compute dtype, dropout probability and seeds."""

    def __init__(self, dt, p: float, training: bool):
        self.dt = dt
        self.p = float(p) if training else 0.0
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if self.p > 0 else 0
        if self.p > 0 and torch.distributed.is_available() and torch.distributed.is_initialized():
            self.seed += 1000003 * torch.distributed.get_rank()       # identically seeded data-parallel ranks draw different masks
        self.site = 0

    def next_seed(self) -> int:
        self.site += 1
        return self.seed + 7919 * self.site


# ----------------------------------------------------------------------------------------------------------------- GEMM helpers
# fp32 products of the text / motion-anchor encoders (activations, gradients and weights stay fp32 there in every mode).  In bf16 training
# mode they run on f16x3 split operands (three f16 MFMA products per K slab, fp32-class result: include/mage_hip.h) instead of the
# exact-fp32 MFMA chain, 3x its rate -- as in generation (mage_model._lin_fp32); precision 'fp32' keeps the exact chain, which is what the
# 1e-4 gradient gates against the oracle are stated for.  _SPLIT32 is set by train_forward / train_backward.
_SPLIT32 = {"sk": 0, "w": {}}
def _f32_branch() -> bool:
    """fp32 branch rows / LayerNorm-output gradients in bf16 mode too (the round-2 form; config.train_f32_branch)."""
    return config.get().train_f32_branch


def _gemm32(a, w, y, *, M, N, K, lda=None, **kw):
    """ops.gemm for fp32 operands; plain products (bias / fp32 residual / QuickGELU only) go through split copies when _SPLIT32 says so."""
    sk = _SPLIT32["sk"]
    lda = K if lda is None else lda
    ldy = kw.pop("ldy", N)
    if (sk and a.dtype == F32 and w.dtype == F32 and K % 64 == 0 and N % 8 == 0 and lda == K and ldy == N and a.dim() == 2 and a.stride(1) == 1
            and w.dim() == 2 and w.is_contiguous() and set(kw) <= {"bias", "residual", "ldr", "act"}
            and kw.get("act", ops.ACT_NONE) in (ops.ACT_NONE, ops.ACT_QUICKGELU) and (kw.get("residual") is None or kw.get("act", 0) == ops.ACT_NONE)):
        key = (w.data_ptr(), tuple(w.shape), w._version)
        ws = _SPLIT32["w"].get(key)
        if ws is None:
            ws = _SPLIT32["w"][key] = (ops.split(w, sk), w)          # keeps w alive: the key is its address
        return ops.gemm(ops.split(a[:M], sk), ws[0], y, M=M, N=N, K=K, lda=2 * K, ldy=N, split_kind=sk, **kw)
    return ops.gemm(a, w, y, M=M, N=N, K=K, lda=lda, ldy=ldy, **kw)


def _gemm_x(a, wT, y, *, M, N, K, lda=None, **kw):
    """y[M, N] = a[M, K] @ wT[N, K]^T (no bias): dX of a Linear, or any plain product."""
    return _gemm32(a, wT, y, M=M, N=N, K=K, lda=lda, **kw)


def _split_plan(M: int, N: int, K: int):
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    S = max(1, min(64, (512 + tiles - 1) // tiles, M // 512 if M >= 1024 else 1))
    Mc = ((M + S - 1) // S + 63) // 64 * 64
    return S, Mc


def _wgrad(dy, x, *, M: int, N: int, K: int, ld_dy: int, ld_x: int, dy_geo: Optional[dict] = None, x_geo: Optional[dict] = None,
           want_bias: bool = True):
    """dW[N, K] = sum_m dy[m, :N]^T x[m, :K] (fp32) and db[N] = sum_m dy[m, :].  dy and x are row sets of the same M logical rows
    (optionally regrouped views: dict(out_w, img_stride, a_off), the [:, 1:] / [:, 0] slices of [B, L, hw, C] streams), same
    dtype (the compute dtype)."""
    assert dy.dtype == x.dtype
    dev, dt = dy.device, dy.dtype
    if (dt == BF16 and dy_geo is None and x_geo is None and N % 256 == 0 and K % 256 == 0 and M >= 4096 and ld_dy % 8 == 0 and ld_x % 8 == 0
            and not config.get().train_wgrad_transpose):
        # the decoder stack's Linear layers: dW = dY^T X straight from the row-major operands (mage_gemm_tn: transposing LDS loads),
        # no dY^T / X^T copies (they were 11.8 ms of a 92 ms step at cfg2)
        return ops.gemm_tn(dy, x, T=M, N=N, K=K, ld_dy=ld_dy, ld_x=ld_x, want_bias=want_bias)
    S, Mc = _split_plan(M, N, K)
    Mp = S * Mc
    dyT = torch.empty(N, Mp, device=dev, dtype=dt)
    db = None
    fused = want_bias and dt == BF16 and N % 8 == 0 and ld_dy % 8 == 0 and Mp % 8 == 0
    if fused:                                       # db in the same pass as the transposed copy (bf16)
        db = ops.transpose_colsum(dy, dyT, M=M, Mp=Mp, C=N, ldx=ld_dy, ldy=Mp, **(dy_geo or {}))
    else:
        ops.transpose(dy, dyT, M=M, Mp=Mp, C=N, ldx=ld_dy, ldy=Mp, **(dy_geo or {}))
    xT = torch.empty(K, Mp, device=dev, dtype=dt)
    ops.transpose(x, xT, M=M, Mp=Mp, C=K, ldx=ld_x, ldy=Mp, **(x_geo or {}))
    part = torch.empty(S, N, K, device=dev, dtype=F32)
    ops.gemm(dyT, xT, part, M=N, N=K, K=Mc, lda=Mp, ldy=K, ldw=Mp, n_split=S, a_split_stride=Mc, w_split_stride=Mc, y_split_stride=N * K)
    if S == 1:
        dW = part[0]
    else:
        dW = ops.sum_partials(part, torch.empty(N, K, device=dev, dtype=F32), stride=N * K, n_part=S, n=N * K)
    if want_bias and not fused:
        db = ops.row_sum(dyT, torch.empty(N, device=dev, dtype=F32), ld=Mp, n=M, rows=N)
    return dW, db


def _to_dt(run: _Run, g32: torch.Tensor, seed: Optional[int] = None) -> torch.Tensor:
    """The fp32 gradient stream as a GEMM operand in the compute dtype, through the dropout mask of the branch it enters."""
    if seed is not None and run.p > 0:
        return ops.dropout(g32, torch.empty(g32.shape, device=g32.device, dtype=run.dt), run.p, seed)
    if run.dt == F32:
        return g32
    return ops.cast(g32, torch.empty(g32.shape, device=g32.device, dtype=run.dt))


def _res_linear(run: _Run, a, d, name, x_old, dt, *, M, N, K, seed, ln=None, cast_to=None):
    """x_new = x_old + dropout(a @ W^T + b): a fresh tensor (x_old is the saved LayerNorm input of the backward pass).
    ln = (gamma, beta, out dtype): also returns LayerNorm(x_new) -- the norm that opens the next branch -- as (x_new, xn); with dropout on,
    the mask-and-add pass and the norm are one launch (mage_dropout_add_layernorm)."""
    w, b = d[name + _sfx(dt)], d.get(name + ".b")
    if run.p == 0:
        x_new = _gemm32(a, w, torch.empty_like(x_old), M=M, N=N, K=K, lda=K, ldy=N, bias=b, residual=x_old, ldr=N)
        if ln is None:
            return x_new if cast_to is None else (x_new, ops.cast(x_new, torch.empty(M, N, device=a.device, dtype=cast_to)))
        return x_new, ops.layernorm(x_new, ln[0], ln[1], torch.empty(M, N, device=a.device, dtype=ln[2]), 1e-5)
    # the branch rows leave the GEMM in the compute dtype (bf16 mode: half the bytes written here and read by the mask-and-add pass;
    # the stream x itself stays fp32).  MAGE_TRAIN_F32_BRANCH=1 keeps fp32 branch rows.
    br = _gemm32(a, w, torch.empty(M, N, device=a.device, dtype=F32 if _f32_branch() else a.dtype), M=M, N=N, K=K, lda=K, ldy=N, bias=b)
    if ln is None:
        if cast_to is not None:                      # the last block: its rows also as the head GEMM's bf16 operand
            xb = torch.empty(M, N, device=a.device, dtype=cast_to)
            return ops.dropout_add(br, x_old, torch.empty_like(x_old), run.p, seed, y_bf16=xb), xb
        return ops.dropout_add(br, x_old, torch.empty_like(x_old), run.p, seed)
    if not config.get().train_emit:
        x_new = ops.dropout_add(br, x_old, torch.empty_like(x_old), run.p, seed)
        return x_new, ops.layernorm(x_new, ln[0], ln[1], torch.empty(M, N, device=a.device, dtype=ln[2]), 1e-5)
    return ops.dropout_add_layernorm(br, x_old, torch.empty_like(x_old), ln[0], ln[1], torch.empty(M, N, device=a.device, dtype=ln[2]), 1e-5,
                                     run.p, seed)


# ----------------------------------------------------------------------------------------------------------------- decoder stack
def _dec_forward(gm, run: _Run, motion, feats, B: int, hh: int, ww: int):
    d = gm._derived.get(gm._build)
    dt, Cc, L, dev = run.dt, gm.model_channels, gm.frames_length, motion.device
    hw = hh * ww
    M, H = B * L * hw, Cc // 32
    x = torch.empty(M, Cc, device=dev, dtype=F32)
    ops.gemm(motion, d["context_linear" + _sfx(dt)], x, M=B * hw, N=Cc, K=gm.context_channels, lda=gm.context_channels, ldy=Cc,
             bias=d["context_linear.b"], out_w=hw, y_img_stride=L * hw, rowadd=d["tpos"], rowadd_div=hw, rowadd_mod=L)
    ops.gemm(feats, d["in_linear" + _sfx(dt)], x, M=B * (L - 1) * hw, N=Cc, K=gm.in_channels, lda=gm.in_channels, ldy=Cc,
             bias=d["in_linear.b"], out_w=(L - 1) * hw, y_img_stride=L * hw, y_off=hw, rowadd=d["tpos"], rowadd_div=hw, rowadd_mod=L)
    blocks = []
    xn1 = xa = None
    for i in range(gm.layers):
        p = f"b{i}"
        axis = i % 3
        if axis == 0:
            geo = dict(n_seq=B * hw, inner=hw, nq=L, nk=L, q_outer_stride=L * hw, q_axis_stride=hw, causal=True)
        elif axis == 1:
            geo = dict(n_seq=B * L * ww, inner=ww, nq=hh, nk=hh, q_outer_stride=hw, q_axis_stride=ww, causal=False)
        else:
            geo = dict(n_seq=B * L * hh, inner=1, nq=ww, nk=ww, q_outer_stride=ww, q_axis_stride=1, causal=False)
        geo.update(kv_outer_stride=geo["q_outer_stride"], kv_axis_stride=geo["q_axis_stride"], n_head=H)
        if xn1 is None:                              # later blocks: written with the previous block's residual add
            xn1 = ops.layernorm(x, d[p + ".ln_1.w"], d[p + ".ln_1.b"], torch.empty(M, Cc, device=dev, dtype=dt), 1e-5)
        qkv = ops.gemm(xn1, d[p + ".in_proj" + _sfx(dt)], torch.empty(M, 3 * Cc, device=dev, dtype=dt), M=M, N=3 * Cc, K=Cc, lda=Cc,
                       ldy=3 * Cc, bias=d[p + ".in_proj.b"])
        ao = torch.empty(M, Cc, device=dev, dtype=dt)
        ops.attention(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], ao, ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc, ldo=Cc, **geo)
        s_attn, s_mlp = run.next_seed(), run.next_seed()
        x1, xn2 = _res_linear(run, ao, d, p + ".out_proj", x, dt, M=M, N=Cc, K=Cc, seed=s_attn, ln=(d[p + ".ln_2.w"], d[p + ".ln_2.b"], dt))
        if dt != F32 and M % 256 == 0 and Cc % 64 == 0 and config.get().train_dual:
            # one launch writes the pre-activation rows (kept for the backward pass) and QuickGELU of them (the next Linear's operand)
            hdn = torch.empty(M, 4 * Cc, device=dev, dtype=dt)
            hpre = ops.gemm(xn2, d[p + ".c_fc" + _sfx(dt)], torch.empty(M, 4 * Cc, device=dev, dtype=dt), M=M, N=4 * Cc, K=Cc, lda=Cc,
                            ldy=4 * Cc, bias=d[p + ".c_fc.b"], act=ops.ACT_QUICKGELU, y2=hdn, ldy2=4 * Cc)
        else:
            hpre = ops.gemm(xn2, d[p + ".c_fc" + _sfx(dt)], torch.empty(M, 4 * Cc, device=dev, dtype=dt), M=M, N=4 * Cc, K=Cc, lda=Cc,
                            ldy=4 * Cc, bias=d[p + ".c_fc.b"])
            hdn = ops.act(hpre, torch.empty_like(hpre), ops.ACT_QUICKGELU)
        nxt = f"b{i + 1}" if i + 1 < gm.layers else None
        if nxt is not None:
            x2, xn_next = _res_linear(run, hdn, d, p + ".c_proj", x1, dt, M=M, N=Cc, K=4 * Cc, seed=s_mlp,
                                      ln=(d[nxt + ".ln_1.w"], d[nxt + ".ln_1.b"], dt))
        else:
            xa = None
            if gm.use_cids and dt == BF16:
                x2, xa = _res_linear(run, hdn, d, p + ".c_proj", x1, dt, M=M, N=Cc, K=4 * Cc, seed=s_mlp, cast_to=dt)
            else:
                x2 = _res_linear(run, hdn, d, p + ".c_proj", x1, dt, M=M, N=Cc, K=4 * Cc, seed=s_mlp)
            xn_next = None
        # hdn is kept for the c_proj weight gradient (recomputing it was one more pass over [M, 4C] per block; 288 GB of HBM)
        blocks.append(dict(x0=x, xn1=xn1, qkv=qkv, ao=ao, x1=x1, xn2=xn2, hpre=hpre, hdn=hdn, geo=geo, s_attn=s_attn, s_mlp=s_mlp))
        x, xn1 = x2, xn_next
    if not gm.use_cids:
        # MAGE+ head (mage_model.py:350-354,387-388): GroupNorm(32) over all L-1 predicted frames of a clip -> SiLU -> Conv3d 1x1x1
        M1 = B * (L - 1) * hw
        st = torch.empty(B, 32, 2, device=dev, dtype=F32)
        y = ops.groupnorm_act(x, d["gn.w"], d["gn.b"], torch.empty(M1, Cc, device=dev, dtype=dt), n_samples=B, rows_per_sample=(L - 1) * hw,
                              sample_stride_rows=L * hw, row_off=hw, groups=32, eps=gm.out[0].eps, act=2, stats=st)
        n8 = d["out.f32"].shape[0]
        pred = ops.gemm(y, d["out" + _sfx(dt)], torch.empty(M1, n8, device=dev, dtype=F32), M=M1, N=n8, K=Cc, lda=Cc, ldy=n8, bias=d["out.b"])
        return pred, dict(blocks=blocks, x_last=x, y=y, gn_stats=st, motion=motion, feats=feats, B=B, hh=hh, ww=ww)
    if dt == F32:
        xa = x
    elif xa is None:
        xa = ops.cast(x, torch.empty(M, Cc, device=dev, dtype=dt))
    Kc = gm.out_channels
    logits = torch.empty(B * (L - 1) * hw, Kc, device=dev, dtype=F32)
    ops.gemm(xa, d["out" + _sfx(dt)], logits, M=B * (L - 1) * hw, N=Kc, K=Cc, lda=Cc, ldy=Kc, bias=d["out.b"], out_w=(L - 1) * hw,
             a_img_stride=L * hw, a_off=hw)
    return logits, dict(blocks=blocks, xa=xa, motion=motion, feats=feats, B=B, hh=hh, ww=ww)


def _emit_next(run, dx, emit):
    """layernorm_bwd's extra output for `emit` = (seed of the branch the updated dx enters next, or None for a plain cast): bf16 mode
    only (the masked / cast copy _to_dt would make in a separate pass); None otherwise."""
    if emit is None or run.dt != BF16 or not config.get().train_emit:
        return None, {}
    seed = emit[0]
    dxb = torch.empty(dx.shape, device=dx.device, dtype=BF16)
    return dxb, dict(dx_bf16=dxb, p=run.p if seed is not None else 0.0, seed=seed or 0)


def _block_mlp_bwd(run, d, p, pre, grads, dx, x1, xn2, hpre, M, Cc, seed, ln_key="ln_2", fc="c_fc", proj="c_proj", act=ops.ACT_QUICKGELU,
                   names=None, hdn=None, dxb=None, emit=None):
    """Backward of x2 = x1 + drop(proj(act(fc(LN(x1))))) given dx = d/dx2 (fp32, updated in place to d/dx1).  dxb: dx as this branch's
    GEMM operand if the caller already holds it; emit = (seed,): returns the updated dx as the NEXT branch's operand (see _emit_next)."""
    dt, dev = run.dt, dx.device
    names = names or {"fc_w": f"{pre}.mlp.c_fc.weight", "fc_b": f"{pre}.mlp.c_fc.bias", "proj_w": f"{pre}.mlp.c_proj.weight",
                      "proj_b": f"{pre}.mlp.c_proj.bias", "ln_w": f"{pre}.ln_2.weight", "ln_b": f"{pre}.ln_2.bias"}
    if dxb is None:
        dxb = _to_dt(run, dx, seed)
    if hdn is None:
        hdn = ops.act(hpre, torch.empty_like(hpre), act)
    grads[names["proj_w"]], grads[names["proj_b"]] = _wgrad(dxb, hdn, M=M, N=Cc, K=4 * Cc, ld_dy=Cc, ld_x=4 * Cc)
    del hdn
    if dt != F32 and act == ops.ACT_QUICKGELU and M % 256 == 0 and Cc % 64 == 0 and config.get().train_dual:
        # d/d(pre-activation) straight from the data-gradient GEMM: its epilogue multiplies the accumulators by QuickGELU'(hpre)
        dh = ops.gemm(dxb, _wt(d, f"{p}.{proj}", dt), torch.empty(M, 4 * Cc, device=dev, dtype=dt), M=M, N=4 * Cc, K=Cc, lda=Cc, ldy=4 * Cc,
                      act=ops.ACT_QUICKGELU_GRAD, y2=hpre, ldy2=4 * Cc)
    else:
        dh = _gemm_x(dxb, _wt(d, f"{p}.{proj}", dt), torch.empty(M, 4 * Cc, device=dev, dtype=dt), M=M, N=4 * Cc, K=Cc)
        ops.act_bwd(hpre, dh, dh, act)
    grads[names["fc_w"]], grads[names["fc_b"]] = _wgrad(dh, xn2, M=M, N=4 * Cc, K=Cc, ld_dy=4 * Cc, ld_x=Cc)
    # d/d(LayerNorm output) in the compute dtype: layernorm_bwd reads it once (its dx stream stays fp32)
    dxn = _gemm_x(dh, _wt(d, f"{p}.{fc}", dt), torch.empty(M, Cc, device=dev, dtype=F32 if _f32_branch() else dt), M=M, N=Cc, K=4 * Cc)
    del dh, dxb
    nxt, kw = _emit_next(run, dx, emit)
    grads[names["ln_w"]], grads[names["ln_b"]] = ops.layernorm_bwd(x1, d[f"{p}.{ln_key}.w"], dxn, dx, eps=1e-5, accumulate=True, **kw)
    return nxt


def _dec_backward(gm, run: _Run, tape, dlogits, grads: Dict[str, torch.Tensor], pre: str = "generate_model"):
    d = gm._derived.get(gm._build)
    dt, Cc, L, dev = run.dt, gm.model_channels, gm.frames_length, dlogits.device
    B, hh, ww = tape["B"], tape["hh"], tape["ww"]
    hw = hh * ww
    M, Kc, M1 = B * L * hw, gm.out_channels, B * (L - 1) * hw
    tail = dict(out_w=(L - 1) * hw, img_stride=L * hw, a_off=hw)            # the x[:, 1:] rows of a [B, L, hw, C] stream
    dx = torch.zeros(M, Cc, device=dev, dtype=F32)
    if gm.use_cids:
        grads[pre + ".out.weight"], grads[pre + ".out.bias"] = _wgrad(dlogits, tape["xa"], M=M1, N=Kc, K=Cc, ld_dy=Kc, ld_x=Cc, x_geo=tail)
        _gemm_x(dlogits, _wt(d, "out", dt), dx, M=M1, N=Cc, K=Kc, out_w=(L - 1) * hw, y_img_stride=L * hw, y_off=hw)
    else:                                                                    # dlogits = d loss / d pred [M1, n8] (padding columns zero)
        n8 = d["out.f32"].shape[0]
        dWo, dbo = _wgrad(dlogits, tape["y"], M=M1, N=n8, K=Cc, ld_dy=n8, ld_x=Cc)
        grads[pre + ".out.2.weight"], grads[pre + ".out.2.bias"] = dWo[:Kc].reshape(Kc, Cc, 1, 1, 1), dbo[:Kc]
        dy = _gemm_x(dlogits, _wt(d, "out", dt), torch.empty(M1, Cc, device=dev, dtype=F32), M=M1, N=Cc, K=n8)
        dg, db, _ = ops.groupnorm_bwd(tape["x_last"], d["gn.w"], d["gn.b"], tape["gn_stats"], dy, dx, n_samples=B, rows_per_sample=(L - 1) * hw,
                                      sample_stride_rows=L * hw, row_off=hw, groups=32, act=2)       # slot 0 rows stay zero
        grads[pre + ".out.0.weight"], grads[pre + ".out.0.bias"] = dg, db
        del dy
    dxb = None                                                               # dx as the next branch's operand, from layernorm_bwd
    for i in reversed(range(gm.layers)):
        p, bp, t = f"b{i}", f"{pre}.blocks.{i}", tape["blocks"][i]
        dxb = _block_mlp_bwd(run, d, p, bp, grads, dx, t["x1"], t["xn2"], t["hpre"], M, Cc, t["s_mlp"], hdn=t.pop("hdn"), dxb=dxb,
                             emit=(t["s_attn"],))
        if dxb is None:
            dxb = _to_dt(run, dx, t["s_attn"])
        grads[bp + ".attn.out_proj.weight"], grads[bp + ".attn.out_proj.bias"] = _wgrad(dxb, t["ao"], M=M, N=Cc, K=Cc, ld_dy=Cc, ld_x=Cc)
        dao = _gemm_x(dxb, _wt(d, p + ".out_proj", dt), torch.empty(M, Cc, device=dev, dtype=dt), M=M, N=Cc, K=Cc)
        qkv = t["qkv"]
        dqkv = torch.empty(M, 3 * Cc, device=dev, dtype=dt)
        ops.attention_bwd(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], dao, dqkv, dqkv[:, Cc:], dqkv[:, 2 * Cc:], ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc,
                          ldo=Cc, ld_dq=3 * Cc, ld_dk=3 * Cc, ld_dv=3 * Cc, **t["geo"])
        grads[bp + ".attn.in_proj_weight"], grads[bp + ".attn.in_proj_bias"] = _wgrad(dqkv, t["xn1"], M=M, N=3 * Cc, K=Cc, ld_dy=3 * Cc,
                                                                                     ld_x=Cc)
        dxn = _gemm_x(dqkv, _wt(d, p + ".in_proj", dt), torch.empty(M, Cc, device=dev, dtype=F32 if _f32_branch() else dt), M=M, N=Cc, K=3 * Cc)
        del dxb, dao, dqkv
        dxb, kw = _emit_next(run, dx, (tape["blocks"][i - 1]["s_mlp"] if i > 0 else None,))
        grads[bp + ".ln_1.weight"], grads[bp + ".ln_1.bias"] = ops.layernorm_bwd(t["x0"], d[p + ".ln_1.w"], dxn, dx, eps=1e-5, accumulate=True, **kw)
        tape["blocks"][i] = None                                             # release the block's activations
    # x_init = [context_linear(motion) | in_linear(feats)] + T_positional_embedding
    tp = ops.group_rowsum(dx, torch.empty(L, Cc, device=dev, dtype=F32), rows=M, C=Cc, div=hw, mod=L)
    grads[pre + ".T_positional_embedding"] = tp.view(L, 1, 1, Cc)
    if dxb is None:
        dxb = _to_dt(run, dx)
    head = dict(out_w=hw, img_stride=L * hw, a_off=0)                        # the x[:, 0] rows
    grads[pre + ".in_linear.weight"], grads[pre + ".in_linear.bias"] = _wgrad(dxb, tape["feats"], M=M1, N=Cc, K=gm.in_channels, ld_dy=Cc,
                                                                             ld_x=gm.in_channels, dy_geo=tail)
    grads[pre + ".context_linear.weight"], grads[pre + ".context_linear.bias"] = _wgrad(dxb, tape["motion"], M=B * hw, N=Cc,
                                                                                       K=gm.context_channels, ld_dy=Cc,
                                                                                       ld_x=gm.context_channels, dy_geo=head)
    dfeats = _gemm_x(dxb, _wt(d, "in_linear", dt), torch.empty(M1, gm.in_channels, device=dev, dtype=F32), M=M1, N=gm.in_channels, K=Cc,
                     out_w=(L - 1) * hw, a_img_stride=L * hw, a_off=hw)
    dmotion = _gemm_x(dxb, _wt(d, "context_linear", dt), torch.empty(B * hw, gm.context_channels, device=dev, dtype=F32), M=B * hw,
                      N=gm.context_channels, K=Cc, out_w=hw, a_img_stride=L * hw, a_off=0)
    return dfeats, dmotion


# ----------------------------------------------------------------------------------------------------------------- frame features
def _conv_flip(model, d, dt):
    """Weight of the input-gradient convolution: Wf[ci, (ky, kx), co] = W[co, (2-ky, 2-kx), ci]."""
    key = "conv.flip" + _sfx(dt)
    if key not in d:
        Cc = model.vision_width
        w = d["conv" + _sfx(dt)].view(Cc, 3, 3, Cc)
        d[key] = w.flip(1, 2).permute(3, 1, 2, 0).reshape(Cc, 9 * Cc).contiguous()
    return d[key]


def _frame_backward(model, run_dt, tok_rows, emb, dfeats, grads, acc, lat_rows=None):
    """feats = conv3x3(emb[tok]) + (H_pos + W_pos): gradients of the conv weight, the positional tables and the token table
    (accumulated into acc['conv'], acc['hwpos'], acc['emb']: the fp32 prologue pass and the decoder pass both land here).
    use_cids=False: emb = Linear(latents) (mage_model.py:583): lat_rows [rows, LD] fp32 instead of tok_rows, the Linear's gradients
    accumulate in acc['emb_lin.w'] [C, LD] / acc['emb_lin.b']."""
    d = model._derived.get(model._build)
    R, Cc, dev = model.image_resolution, model.vision_width, dfeats.device
    hw = R * R
    rows = dfeats.shape[0]
    n_img = rows // hw
    hp = ops.group_rowsum(dfeats, torch.empty(hw, Cc, device=dev, dtype=F32), rows=rows, C=Cc, div=1, mod=hw)
    acc["hwpos"] = hp if acc.get("hwpos") is None else acc["hwpos"] + hp
    P = R + 2
    taps = run_dt == BF16 and Cc % 256 == 0 and rows % 256 == 0 and config.get().train_taps
    if taps:
        # dfeats as bf16 rows in the interior of a zero-padded (R+2) x (R+2) frame buffer (border written once): its transposed
        # convolution below is then the padded-taps form of mage_gemm (8-phase kernel, 1.4 -> 0.9 ms at cfg2), as in _frame_features
        key = ("dfeats", n_img, str(dev))
        pad = model._pad_frames.get(key)
        if pad is None:
            pad = model._pad_frames[key] = torch.zeros((n_img * P * P + 1) * Cc, device=dev, dtype=run_dt).view(-1, Cc)
        pad[:n_img * P * P].view(n_img, P, P, Cc)[:, 1:R + 1, 1:R + 1].copy_(dfeats.view(n_img, R, R, Cc))       # cast + placement
        dfe, dfe_geo = pad, dict(out_h=R, out_w=R, in_h=P, in_w=P, img_stride=P * P, dy=1, dx=1)
    else:
        dfe = dfeats if run_dt == F32 else ops.cast(dfeats, torch.empty(rows, Cc, device=dev, dtype=run_dt))
        dfe_geo = {}
    # dW[co, tap, ci] = sum_p dfeats[p, co] * emb[shift_tap(p), ci]: nine shifted transposes stacked, one split-K GEMM
    S, Mc = _split_plan(rows, Cc, 9 * Cc)
    Mp = S * Mc
    dyT = torch.empty(Cc, Mp, device=dev, dtype=run_dt)
    ops.transpose(dfe, dyT, M=rows, Mp=Mp, C=Cc, ldx=Cc, ldy=Mp, **dfe_geo)
    xT = torch.empty(9 * Cc, Mp, device=dev, dtype=run_dt)
    for ky in range(3):
        for kx in range(3):
            ops.transpose(emb, xT, M=rows, Mp=Mp, C=Cc, ldx=Cc, ldy=Mp, y_row0=(ky * 3 + kx) * Cc, out_h=R, out_w=R, in_h=R, in_w=R,
                          img_stride=hw, dy=ky - 1, dx=kx - 1)
    part = torch.empty(S, Cc, 9 * Cc, device=dev, dtype=F32)
    ops.gemm(dyT, xT, part, M=Cc, N=9 * Cc, K=Mc, lda=Mp, ldy=9 * Cc, ldw=Mp, n_split=S, a_split_stride=Mc, w_split_stride=Mc,
             y_split_stride=Cc * 9 * Cc)
    dW = part[0] if S == 1 else ops.sum_partials(part, torch.empty(Cc, 9 * Cc, device=dev, dtype=F32), stride=Cc * 9 * Cc, n_part=S,
                                                 n=Cc * 9 * Cc)
    acc["conv"] = dW if acc.get("conv") is None else acc["conv"] + dW
    del dyT, xT, part
    # d emb = the transposed convolution of dfeats
    if taps:
        demb = ops.gemm(dfe, _conv_flip(model, d, run_dt), torch.empty(rows, Cc, device=dev, dtype=F32), M=rows, N=Cc, K=9 * Cc, lda=Cc, ldy=Cc,
                        out_h=R, out_w=R, in_h=P, in_w=P, a_img_stride=P * P, taps_h=3, taps_w=3, cin=Cc, stride=1, dy0=0, dx0=0)
    else:
        demb = VectorQuantizedVAE._conv(dfe, _conv_flip(model, d, run_dt), torch.empty(rows, Cc, device=dev, dtype=F32), n_img=n_img, H=R, W=R,
                                        cin=Cc, cout=Cc, k=3)
    if lat_rows is not None:
        LD = lat_rows.shape[1]
        dW, db = _wgrad(demb, lat_rows, M=rows, N=Cc, K=LD, ld_dy=Cc, ld_x=LD)
        acc["emb_lin.w"] = dW if acc.get("emb_lin.w") is None else acc["emb_lin.w"] + dW
        acc["emb_lin.b"] = db if acc.get("emb_lin.b") is None else acc["emb_lin.b"] + db
        return
    if acc.get("emb") is None:
        acc["emb"] = torch.zeros(model.codebook_size, Cc, device=dev, dtype=F32)
    ops.embedding_bwd(tok_rows, demb, acc["emb"])


# ----------------------------------------------------------------------------------------------------------------- MA encoder
def _ma_forward(ma, run: _Run, q, kv, B: int, nq: int, nk: int):
    """MAEncoder (one or more TransformerBlocks, the MAGE variant mage_model.py:92), batch-first rows, fp32."""
    d = ma._derived.get(ma._build)
    Cc, dev, H = ma.d_model, q.device, ma.d_model // 32
    x = q
    layers = []
    for i in range(ma.layers):
        p = f"b{i}"
        w, b = d[p + ".in_proj.f32"], d[p + ".in_proj.b"]
        qin, kvin = x, kv
        if ma.mage_plus:                                                     # the ln_q / ln_kv line of mage_model.py:93 (MAGE+)
            qin = ops.layernorm(x, d[p + ".ln_q.w"], d[p + ".ln_q.b"], torch.empty_like(x), 1e-5)
            kvin = ops.layernorm(kv, d[p + ".ln_kv.w"], d[p + ".ln_kv.b"], torch.empty_like(kv), 1e-5)
        qp = _gemm32(qin, w[:Cc], torch.empty(B * nq, Cc, device=dev, dtype=F32), M=B * nq, N=Cc, K=Cc, lda=Cc, ldy=Cc, bias=b[:Cc])
        kvp = _gemm32(kvin, w[Cc:], torch.empty(B * nk, 2 * Cc, device=dev, dtype=F32), M=B * nk, N=2 * Cc, K=Cc, lda=Cc, ldy=2 * Cc, bias=b[Cc:])
        geo = dict(n_seq=B, inner=1, nq=nq, nk=nk, n_head=H, q_outer_stride=nq, q_axis_stride=1, kv_outer_stride=nk, kv_axis_stride=1)
        ao = torch.empty(B * nq, Cc, device=dev, dtype=F32)
        ops.attention(qp, kvp, kvp[:, Cc:], ao, ldq=Cc, ldk=2 * Cc, ldv=2 * Cc, ldo=Cc, **geo)
        s_attn, s_mlp = run.next_seed(), run.next_seed()
        x1 = _res_linear(run, ao, d, p + ".out_proj", x, F32, M=B * nq, N=Cc, K=Cc, seed=s_attn)
        xn = ops.layernorm(x1, d[p + ".ln_2.w"], d[p + ".ln_2.b"], torch.empty_like(x1), 1e-5)
        hpre = _gemm32(xn, d[p + ".c_fc.f32"], torch.empty(B * nq, 4 * Cc, device=dev, dtype=F32), M=B * nq, N=4 * Cc, K=Cc, lda=Cc,
                       ldy=4 * Cc, bias=d[p + ".c_fc.b"])
        hdn = ops.act(hpre, torch.empty_like(hpre), ops.ACT_QUICKGELU)
        x2 = _res_linear(run, hdn, d, p + ".c_proj", x1, F32, M=B * nq, N=Cc, K=4 * Cc, seed=s_mlp)
        layers.append(dict(x0=x, qin=qin, kvin=kvin, qp=qp, kvp=kvp, ao=ao, x1=x1, xn=xn, hpre=hpre, geo=geo, s_attn=s_attn, s_mlp=s_mlp))
        x = x2
    return x, dict(layers=layers, kv=kv, B=B, nq=nq, nk=nk)


def _ma_backward(ma, run32: _Run, tape, dx, grads, pre: str = "ma_encoder"):
    d = ma._derived.get(ma._build)
    Cc, dev = ma.d_model, dx.device
    B, nq, nk, kv = tape["B"], tape["nq"], tape["nk"], tape["kv"]
    dkv_total = None
    dx = dx.clone()
    for i in reversed(range(ma.layers)):
        p, bp, t = f"b{i}", f"{pre}.blocks.{i}", tape["layers"][i]
        _block_mlp_bwd(run32, d, p, bp, grads, dx, t["x1"], t["xn"], t["hpre"], B * nq, Cc, t["s_mlp"])
        dbr = _to_dt(run32, dx, t["s_attn"])
        grads[bp + ".attn.out_proj.weight"], grads[bp + ".attn.out_proj.bias"] = _wgrad(dbr, t["ao"], M=B * nq, N=Cc, K=Cc, ld_dy=Cc, ld_x=Cc)
        dao = _gemm_x(dbr, _wt(d, p + ".out_proj", F32), torch.empty(B * nq, Cc, device=dev, dtype=F32), M=B * nq, N=Cc, K=Cc)
        dqp = torch.empty(B * nq, Cc, device=dev, dtype=F32)
        dkvp = torch.empty(B * nk, 2 * Cc, device=dev, dtype=F32)
        kvp = t["kvp"]
        ops.attention_bwd(t["qp"], kvp, kvp[:, Cc:], dao, dqp, dkvp, dkvp[:, Cc:], ldq=Cc, ldk=2 * Cc, ldv=2 * Cc, ldo=Cc, ld_dq=Cc,
                          ld_dk=2 * Cc, ld_dv=2 * Cc, **t["geo"])
        dWq, dbq = _wgrad(dqp, t["qin"], M=B * nq, N=Cc, K=Cc, ld_dy=Cc, ld_x=Cc)
        dWkv, dbkv = _wgrad(dkvp, t["kvin"], M=B * nk, N=2 * Cc, K=Cc, ld_dy=2 * Cc, ld_x=Cc)
        grads[bp + ".attn.in_proj_weight"] = torch.cat([dWq, dWkv], 0)                      # [3C, C] assembly (layout plumbing)
        grads[bp + ".attn.in_proj_bias"] = torch.cat([dbq, dbkv], 0)
        wT = _wt(d, p + ".in_proj", F32)                                                    # [C, 3C]
        wqT, wkvT = wT[:, :Cc].contiguous(), wT[:, Cc:].contiguous()
        if ma.mage_plus:
            dqin = _gemm_x(dqp, wqT, torch.empty(B * nq, Cc, device=dev, dtype=F32), M=B * nq, N=Cc, K=Cc)
            grads[bp + ".ln_q.weight"], grads[bp + ".ln_q.bias"] = ops.layernorm_bwd(t["x0"], d[p + ".ln_q.w"], dqin, dx, eps=1e-5, accumulate=True)
            dkvin = _gemm_x(dkvp, wkvT, torch.empty(B * nk, Cc, device=dev, dtype=F32), M=B * nk, N=Cc, K=2 * Cc)
            dkv = torch.empty_like(dkvin)
            grads[bp + ".ln_kv.weight"], grads[bp + ".ln_kv.bias"] = ops.layernorm_bwd(kv, d[p + ".ln_kv.w"], dkvin, dkv, eps=1e-5,
                                                                                      accumulate=False)
        else:
            dx = _gemm_x(dqp, wqT, torch.empty(B * nq, Cc, device=dev, dtype=F32), M=B * nq, N=Cc, K=Cc, residual=dx, ldr=Cc)
            dkv = _gemm_x(dkvp, wkvT, torch.empty(B * nk, Cc, device=dev, dtype=F32), M=B * nk, N=Cc, K=2 * Cc)
            # ln_q / ln_kv exist in the checkpoint but are not applied by MAGE (mage_model.py:92): zero gradients
        dkv_total = dkv if dkv_total is None else dkv_total + dkv
    return dx, dkv_total


# ----------------------------------------------------------------------------------------------------------------- text encoder
def _text_forward(te, run: _Run, text):
    d = te._derived.get(te._build)
    B, S = text.shape
    Wd, dev, H = te.transformer_width, text.device, te.transformer_width // 32
    ids = text.to(torch.int64).contiguous()
    keep = ids != te.padding_idx
    kv_len = keep.sum(-1).to(torch.int32).contiguous()
    keepf = keep.reshape(-1).to(F32).contiguous()
    e = ops.embedding(ids, d["tok"], torch.empty(B * S, Wd, device=dev, dtype=F32))
    ops.row_affine(e, None, d["pos"], div=1, mod=S)
    x = ops.layernorm(e, d["layer_norm.w"], d["layer_norm.b"], torch.empty_like(e), te.layer_norm.eps)
    s_emb = run.next_seed()
    if run.p > 0:
        x = ops.dropout(x, torch.empty_like(x), run.p, s_emb)
    ops.row_affine(x, keepf, None)
    geo = dict(n_seq=B, inner=1, nq=S, nk=S, n_head=H, q_outer_stride=S, q_axis_stride=1, kv_outer_stride=S, kv_axis_stride=1,
               kv_len=kv_len, kv_len_div=1)
    layers = []
    for i in range(te.transformer_layers):
        p = f"l{i}"
        qkv = _gemm32(x, d[p + ".in_proj.f32"], torch.empty(B * S, 3 * Wd, device=dev, dtype=F32), M=B * S, N=3 * Wd, K=Wd, lda=Wd,
                      ldy=3 * Wd, bias=d[p + ".in_proj.b"])
        ao = torch.empty(B * S, Wd, device=dev, dtype=F32)
        # nn.TransformerEncoderLayer(dropout=p) also drops attention PROBABILITIES in train() (mage_model.py:193-199)
        sa_seed = run.next_seed()
        ops.attention(qkv, qkv[:, Wd:], qkv[:, 2 * Wd:], ao, ldq=3 * Wd, ldk=3 * Wd, ldv=3 * Wd, ldo=Wd, drop_p=run.p, drop_seed=sa_seed, **geo)
        s1_seed, sh_seed, s2_seed = run.next_seed(), run.next_seed(), run.next_seed()
        s1 = _res_linear(run, ao, d, p + ".out_proj", x, F32, M=B * S, N=Wd, K=Wd, seed=s1_seed)
        x1 = ops.layernorm(s1, d[p + ".norm1.w"], d[p + ".norm1.b"], torch.empty_like(s1), d[p + ".norm1.eps"])
        hpre = _gemm32(x1, d[p + ".fc1.f32"], torch.empty(B * S, 4 * Wd, device=dev, dtype=F32), M=B * S, N=4 * Wd, K=Wd, lda=Wd,
                       ldy=4 * Wd, bias=d[p + ".fc1.b"])
        hdn = ops.act(hpre, torch.empty_like(hpre), ops.ACT_GELU_ERF)
        if run.p > 0:
            hdn = ops.dropout(hdn, torch.empty_like(hdn), run.p, sh_seed)
        s2 = _res_linear(run, hdn, d, p + ".fc2", x1, F32, M=B * S, N=Wd, K=4 * Wd, seed=s2_seed)
        x2 = ops.layernorm(s2, d[p + ".norm2.w"], d[p + ".norm2.b"], torch.empty_like(s2), d[p + ".norm2.eps"])
        layers.append(dict(x_in=x, qkv=qkv, ao=ao, s1=s1, x1=x1, hpre=hpre, s2=s2, seeds=(s1_seed, sh_seed, s2_seed), sa_seed=sa_seed))
        x = x2
    xf = ops.layernorm(x, d["ln_text_final.w"], d["ln_text_final.b"], torch.empty_like(x), te.ln_text_final.eps)
    out = _gemm32(xf, d["proj.f32"], torch.empty(B * S, te.output_dim, device=dev, dtype=F32), M=B * S, N=te.output_dim, K=Wd, lda=Wd,
                   ldy=te.output_dim, bias=d["proj.b"])
    return out, dict(ids=ids, keepf=keepf, e=e, s_emb=s_emb, layers=layers, x_last=x, xf=xf, geo=geo, B=B, S=S)


def _text_backward(te, run32: _Run, tape, dout, grads, pre: str = "text_encoder"):
    d = te._derived.get(te._build)
    B, S = tape["B"], tape["S"]
    Wd, dev, R = te.transformer_width, dout.device, tape["B"] * tape["S"]
    grads[pre + ".text_projection.weight"], grads[pre + ".text_projection.bias"] = _wgrad(dout, tape["xf"], M=R, N=te.output_dim, K=Wd,
                                                                                         ld_dy=te.output_dim, ld_x=Wd)
    dxf = _gemm_x(dout, _wt(d, "proj", F32), torch.empty(R, Wd, device=dev, dtype=F32), M=R, N=Wd, K=te.output_dim)
    dx = torch.empty(R, Wd, device=dev, dtype=F32)
    grads[pre + ".ln_text_final.weight"], grads[pre + ".ln_text_final.bias"] = ops.layernorm_bwd(tape["x_last"], d["ln_text_final.w"], dxf,
                                                                                                dx, eps=te.ln_text_final.eps,
                                                                                                accumulate=False)
    for i in reversed(range(te.transformer_layers)):
        p, lp, t = f"l{i}", f"{pre}.transformer.layers.{i}", tape["layers"][i]
        s1_seed, sh_seed, s2_seed = t["seeds"]
        ds2 = torch.empty(R, Wd, device=dev, dtype=F32)
        grads[lp + ".norm2.weight"], grads[lp + ".norm2.bias"] = ops.layernorm_bwd(t["s2"], d[p + ".norm2.w"], dx, ds2, eps=d[p + ".norm2.eps"],
                                                                                  accumulate=False)
        dbr = _to_dt(run32, ds2, s2_seed)
        hdn = ops.act(t["hpre"], torch.empty_like(t["hpre"]), ops.ACT_GELU_ERF)
        if run32.p > 0:
            hdn = ops.dropout(hdn, torch.empty_like(hdn), run32.p, sh_seed)
        grads[lp + ".linear2.weight"], grads[lp + ".linear2.bias"] = _wgrad(dbr, hdn, M=R, N=Wd, K=4 * Wd, ld_dy=Wd, ld_x=4 * Wd)
        dh = _gemm_x(dbr, _wt(d, p + ".fc2", F32), torch.empty(R, 4 * Wd, device=dev, dtype=F32), M=R, N=4 * Wd, K=Wd)
        if run32.p > 0:
            dh = ops.dropout(dh, torch.empty_like(dh), run32.p, sh_seed)
        ops.act_bwd(t["hpre"], dh, dh, ops.ACT_GELU_ERF)
        grads[lp + ".linear1.weight"], grads[lp + ".linear1.bias"] = _wgrad(dh, t["x1"], M=R, N=4 * Wd, K=Wd, ld_dy=4 * Wd, ld_x=Wd)
        dx1 = _gemm_x(dh, _wt(d, p + ".fc1", F32), torch.empty(R, Wd, device=dev, dtype=F32), M=R, N=Wd, K=4 * Wd, residual=ds2, ldr=Wd)
        ds1 = torch.empty(R, Wd, device=dev, dtype=F32)
        grads[lp + ".norm1.weight"], grads[lp + ".norm1.bias"] = ops.layernorm_bwd(t["s1"], d[p + ".norm1.w"], dx1, ds1, eps=d[p + ".norm1.eps"],
                                                                                  accumulate=False)
        dbr = _to_dt(run32, ds1, s1_seed)
        grads[lp + ".self_attn.out_proj.weight"], grads[lp + ".self_attn.out_proj.bias"] = _wgrad(dbr, t["ao"], M=R, N=Wd, K=Wd, ld_dy=Wd, ld_x=Wd)
        dao = _gemm_x(dbr, _wt(d, p + ".out_proj", F32), torch.empty(R, Wd, device=dev, dtype=F32), M=R, N=Wd, K=Wd)
        qkv = t["qkv"]
        dqkv = torch.empty(R, 3 * Wd, device=dev, dtype=F32)
        ops.attention_bwd(qkv, qkv[:, Wd:], qkv[:, 2 * Wd:], dao, dqkv, dqkv[:, Wd:], dqkv[:, 2 * Wd:], ldq=3 * Wd, ldk=3 * Wd, ldv=3 * Wd,
                          ldo=Wd, ld_dq=3 * Wd, ld_dk=3 * Wd, ld_dv=3 * Wd, drop_p=run32.p, drop_seed=t["sa_seed"], **tape["geo"])
        grads[lp + ".self_attn.in_proj_weight"], grads[lp + ".self_attn.in_proj_bias"] = _wgrad(dqkv, t["x_in"], M=R, N=3 * Wd, K=Wd,
                                                                                               ld_dy=3 * Wd, ld_x=Wd)
        dx = _gemm_x(dqkv, _wt(d, p + ".in_proj", F32), torch.empty(R, Wd, device=dev, dtype=F32), M=R, N=Wd, K=3 * Wd, residual=ds1, ldr=Wd)
    ops.row_affine(dx, tape["keepf"], None)                                   # x = dropout(LN(e)) * keep
    if run32.p > 0:
        dx = ops.dropout(dx, torch.empty_like(dx), run32.p, tape["s_emb"])
    de = torch.empty(R, Wd, device=dev, dtype=F32)
    grads[pre + ".layer_norm.weight"], grads[pre + ".layer_norm.bias"] = ops.layernorm_bwd(tape["e"], d["layer_norm.w"], dx, de,
                                                                                          eps=te.layer_norm.eps, accumulate=False)
    gt = torch.zeros(te.vocab_size, Wd, device=dev, dtype=F32)
    ops.embedding_bwd(tape["ids"].reshape(-1), de, gt, padding_idx=te.padding_idx if te.padding_idx is not None else -1)
    grads[pre + ".token_embedding.weight"] = gt
    gp = torch.zeros(te.context_length, Wd, device=dev, dtype=F32)
    ops.group_rowsum(de, gp, rows=R, C=Wd, div=1, mod=S)                      # rows 0..S-1 of the positions table
    grads[pre + ".positions.weight"] = gp


# ----------------------------------------------------------------------------------------------------------------- whole model
def train_forward(model, batch):
    """Teacher-forced pass of MAGE.forward (mage_model.py:575-639) with the activations the backward pass needs.
    Returns (loss 0-dim fp32 tensor, tape)."""
    images = batch["images"]
    B = images.shape[0]
    R, L, Cc = model.image_resolution, model.frames_length, model.vision_width
    hw = R * R
    dt = model._dt()
    if dt == torch.float16:
        raise ValueError("precision 'f16' is a generation mode (forward values, autoregressive_generate): the backward kernels are bf16 / fp32 -- "
                         "train with set_precision('bf16') or 'fp32'")
    run = _Run(dt, model.dropout, model.training)
    run32 = _Run(F32, model.dropout, model.training)
    run32.seed, run32.p = run.seed, run.p
    run32.site = 1 << 20            # its own stream of dropout sites: the decoder's and the encoders' layers never share a seed
    run32.sk = ops.F16X3 if (dt != F32 and config.get().train_enc_split) else 0
    _SPLIT32["sk"], _SPLIT32["w"] = run32.sk, {}
    d = model._derived.get(model._build)
    dev = images.device
    tok = tok_in = tok0 = lat_all = lat_in = lat0 = None
    if model.use_cids:
        tok = model.first_stage_encode(images).reshape(B, -1, hw)                     # frozen first stage: no gradient
        tok_in = tok[:, :L - 1].contiguous()
        tok0 = tok[:, 0].contiguous()
        # frame features for the decoder (compute dtype) and, as the inference prologue does, frame 0's in fp32 for the MA encoder
        emb = ops.embedding(tok_in.reshape(-1), d["emb"], torch.empty(B * (L - 1) * hw, Cc, device=dev, dtype=dt))
        emb0 = ops.embedding(tok0.reshape(-1), d["emb"], torch.empty(B * hw, Cc, device=dev, dtype=F32))
    else:
        # MAGE+ (mage_model.py:579,583): latents of ALL frames from the external first stage as zero-padded rows [., LD], token
        # embedding = Linear(embed_dim -> C)
        E, LD = model.first_stage_model.embed_dim, 8
        lat = model.first_stage_encode(images)                                        # [B, Lb, E, h, w]
        Lb = lat.shape[1]
        lat_all = torch.zeros(B, Lb, hw, LD, device=dev, dtype=F32)
        lat_all[..., :E] = lat.permute(0, 1, 3, 4, 2).reshape(B, Lb, hw, E).float()   # layout plumbing (channels last)
        lat_in = lat_all[:, :L - 1].contiguous().view(-1, LD)
        lat0 = lat_all[:, 0].contiguous().view(-1, LD)
        lin = dict(N=Cc, K=E, lda=LD, ldy=Cc, bias=d["emb_lin.b"])
        emb = ops.gemm(lat_in, d["emb_lin.w"], torch.empty(B * (L - 1) * hw, Cc, device=dev, dtype=dt), M=B * (L - 1) * hw, **lin)
        emb0 = ops.gemm(lat0, d["emb_lin.w"], torch.empty(B * hw, Cc, device=dev, dtype=F32), M=B * hw, **lin)
    if model.use_cids and dt == BF16 and Cc % 64 == 0 and config.get().train_taps:
        # the generation path's padded-taps convolution (embedding rows written into a zero-padded frame buffer: the 8-phase kernel
        # instead of the per-lane gather, 1.45 -> 0.9 ms at cfg2); emb (plain rows) stays the backward pass's operand
        feats = model._frame_features(tok_in, dt)
    else:
        feats = VectorQuantizedVAE._conv(emb, d["conv" + _sfx(dt)], torch.empty_like(emb), n_img=B * (L - 1), H=R, W=R, cin=Cc, cout=Cc, k=3,
                                         rowadd=d["hwpos"], rowadd_div=1, rowadd_mod=hw)
    first = VectorQuantizedVAE._conv(emb0, d["conv.f32"], torch.empty_like(emb0), n_img=B, H=R, W=R, cin=Cc, cout=Cc, k=3,
                                     rowadd=d["hwpos"], rowadd_div=1, rowadd_mod=hw)
    txt, t_text = _text_forward(model.text_encoder, run32, batch["text"])
    S = batch["text"].shape[1]
    ma, t_ma = _ma_forward(model.ma_encoder, run32, first, txt, B, hw, S)
    t_rand = None
    if model.randomness:                                                     # :601-609: ADAIN modulation by the reparameterised video prior
        from . import mage_train_prior
        ma, t_rand = mage_train_prior.rand_forward(model, batch, tok, ma, B, lat_rows=None if model.use_cids else lat_all.view(-1, 8),
                                                   L=0 if model.use_cids else lat_all.shape[1])
    speed = None
    if "speed" in batch:
        speed = batch["speed"].float().contiguous()
        ma = ma.clone()
        ops.add_scaled_rowvec(ma, speed, d["speed"], B=B, P=hw, Cc=Cc)
    ma_dt = ma if dt == F32 else ma.to(dt)
    logits, t_dec = _dec_forward(model.generate_model, run, ma_dt, feats, B, R, R)
    if model.use_cids:
        target = tok[:, 1:L].reshape(-1).contiguous()
        recon = ops.cross_entropy(logits, target)                                                            # :618
    else:
        target = lat_all[:, 1:L].contiguous().view(-1, 8)
        recon = ops.mse(logits, target, rows=B * (L - 1) * hw, cols=model.first_stage_model.embed_dim, lda=logits.shape[1], ldb=8)   # :620
        model.last_logits = logits
    loss, parts = recon, {"prediction": recon.item()}
    beta = alpha = 0.0
    if model.randomness:                                                     # :622-632 ([B]-element reductions and scalars)
        kl = -0.5 * t_rand["kl_sum"].mean()
        parts["kl_loss"] = kl.item()
        if model.auto_beta:
            model.beta, _ = model.PID.pid(model.KL_loss, parts["kl_loss"])
            parts["beta"] = model.beta
            beta = float(model.beta)
            loss = recon + beta * kl
        else:
            if speed is None:
                raise KeyError("MAGE.forward with randomness=True and auto_beta=False needs batch['speed'] (mage_model.py:631)")
            beta, alpha = float(model.beta), float(model.alpha)
            l2 = (speed ** 2).mean() * (d["speed"] ** 2).sum()
            loss = recon + beta * kl + alpha * l2
    parts["final_loss"] = loss.item()
    tape = dict(run=run, run32=run32, tok_in=tok_in, tok0=tok0, lat_in=lat_in, lat0=lat0, emb=emb, emb0=emb0, text=t_text, ma=t_ma, dec=t_dec,
                logits=logits, target=target, speed=speed, B=B, rand=t_rand, beta=beta, alpha=alpha, parts=parts)
    return loss, tape


def train_backward(model, tape, grad_out: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Gradients of every trainable parameter, keyed by state_dict name (fp32, the parameter's shape)."""
    run, run32 = tape["run"], tape["run32"]
    _SPLIT32["sk"] = getattr(run32, "sk", 0)
    dt = run.dt
    R, L, Cc, B = model.image_resolution, model.frames_length, model.vision_width, tape["B"]
    hw = R * R
    dev = tape["logits"].device
    grads: Dict[str, torch.Tensor] = {}
    gout = grad_out.detach().to(device=dev, dtype=F32).reshape(1).contiguous()
    if model.use_cids:
        dlogits = ops.cross_entropy_bwd(tape["logits"], tape["target"], gout, torch.empty(tape["logits"].shape, device=dev, dtype=dt))
    else:
        pred = tape["logits"]
        dlogits = _to_dt(run, ops.mse_bwd(pred, tape["target"], gout, rows=pred.shape[0], cols=model.first_stage_model.embed_dim,
                                          lda=pred.shape[1], ldb=8))
    tape["logits"] = None
    dfeats, dma = _dec_backward(model.generate_model, run, tape["dec"], dlogits, grads)
    del dlogits
    acc: Dict[str, Optional[torch.Tensor]] = {}
    _frame_backward(model, dt, None if tape["tok_in"] is None else tape["tok_in"].reshape(-1), tape["emb"], dfeats, grads, acc,
                    lat_rows=tape["lat_in"])
    del dfeats
    if tape["speed"] is not None:                                            # ma += speed_b * speed_embedding  (:666-668)
        gs = ops.group_rowsum(dma, torch.empty(1, Cc, device=dev, dtype=F32), rows=B * hw, C=Cc, div=1, mod=1, row_scale=tape["speed"],
                              row_scale_div=hw)
        if tape["alpha"] != 0.0:                                             # alpha * mean_b(speed_b^2) * |speed_embedding|^2  (:631)
            d = model._derived.get(model._build)
            gs = gs + (2.0 * tape["alpha"]) * gout * (tape["speed"] ** 2).mean() * d["speed"].view(1, Cc)
        grads["speed_embedding"] = gs
    if tape["rand"] is not None:
        from . import mage_train_prior
        kl_coef = (gout * (tape["beta"] / B)).contiguous()
        dma = mage_train_prior.rand_backward(model, tape["rand"], dma, kl_coef, grads, acc)
        tape["rand"] = None
    dfirst, dtxt = _ma_backward(model.ma_encoder, run32, tape["ma"], dma, grads)
    _text_backward(model.text_encoder, run32, tape["text"], dtxt, grads)
    _frame_backward(model, F32, None if tape["tok0"] is None else tape["tok0"].reshape(-1), tape["emb0"], dfirst, grads, acc,
                    lat_rows=tape["lat0"])
    if model.use_cids:
        grads["visual_token_embedding.weight"] = acc["emb"]
    else:
        E = model.first_stage_model.embed_dim
        grads["visual_token_embedding.weight"] = acc["emb_lin.w"][:, :E].contiguous()
        grads["visual_token_embedding.bias"] = acc["emb_lin.b"]
    grads["conv.0.weight"] = acc["conv"].view(Cc, 3, 3, Cc).permute(0, 3, 1, 2).contiguous()       # [Cout, Cin, kh, kw] layout
    hp = acc["hwpos"].contiguous()                                           # [R*R, C] -> H table (sum over w), W table (sum over h)
    grads["H_positional_embedding"] = ops.group_rowsum(hp, torch.empty(R, Cc, device=dev, dtype=F32), rows=hw, C=Cc, div=R,
                                                       mod=R).view(1, R, 1, Cc)
    grads["W_positional_embedding"] = ops.group_rowsum(hp, torch.empty(R, Cc, device=dev, dtype=F32), rows=hw, C=Cc, div=1,
                                                       mod=R).view(1, 1, R, Cc)
    return grads


def trainable_names(model):
    return [n for n, p in model.named_parameters() if p.requires_grad]


class MageLossFn(torch.autograd.Function):
    """loss = MageLossFn.apply(model, batch, names, *params): autograd sees one node whose inputs are the trainable parameters."""

    @staticmethod
    def forward(ctx, model, batch, names, *params):
        with torch.no_grad():
            loss, tape = train_forward(model, batch)
        model._last_train_parts = tape["parts"]                              # the reference's loss_dict values (without the prefix)
        ctx.model, ctx.tape, ctx.names, ctx.shapes = model, tape, names, [p.shape for p in params]
        ctx.devices = [p.device for p in params]
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.tape is None:
            raise RuntimeError("MAGE training graph: backward through the same forward a second time (activations were released)")
        with torch.no_grad(), torch.cuda.device(ctx.devices[0]):
            grads = train_backward(ctx.model, ctx.tape, grad_out)
        ctx.tape = None
        out = []
        for n, shp, dev in zip(ctx.names, ctx.shapes, ctx.devices):
            g = grads.get(n)
            out.append(g.reshape(shp) if g is not None else torch.zeros(shp, device=dev, dtype=F32))     # ln_q / ln_kv: unused -> 0
        return (None, None, None, *out)
