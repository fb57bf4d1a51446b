"""Training path of MAGE.forward's randomness branch (mage_model.py:601-609,622-632) on the HIP kernels.

``randomness=True`` (all four configs the reference ships) adds to the teacher-forced pass

    prior  = conv3d(embeddings of ALL frames)            four BasicBlocks: Conv3d 3x3x3 (temporal stride 2) + GroupNorm(16) + ReLU
    mu, logvar = conv_mu2(prior), conv_var2(prior);      z = eps * exp(logvar / 2) + mu
    ma     = ADAIN2D(ma, conv_d2(z))                     InstanceNorm(ma) * conv_mu(y) + conv_var(y)
    loss   = recon + beta * KL(mu, logvar) [+ alpha * l2(speed embedding)]

Forward = the inference kernels (MAGE._video_prior with a tape, the implicit-GEMM 3x3 convolutions, mage_adain, mage_reparam_kl);
backward = the same GEMM kernel on flipped / transposed weights for the input gradients (a Conv3d's temporal taps are three
accumulating launches into the zero-padded frame buffer its forward read from), transposes + one split-K GEMM per weight gradient,
and mage_groupnorm_bwd / mage_adain_bwd / mage_reparam_kl_bwd (csrc/train.hip).  In bf16 training the Conv3d video prior (the bulk of
the branch: 36 temporal-tap convolutions over all frames, forward and twice backward) stores its activations and weights in bf16 for
the matrix cores (convolution outputs, GroupNorm statistics and every gradient stream stay fp32); the per-clip 3x3 heads and ADAIN are
fp32 in both modes, like the once-per-clip prologue of the inference path.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops
from .mage_train import _split_plan, _wgrad
from .vqvae_model import VectorQuantizedVAE

F32 = torch.float32
_conv = VectorQuantizedVAE._conv

__all__ = ["rand_forward", "rand_backward"]


def _flip(d: Dict[str, torch.Tensor], key: str, cin: int, cout: int, dt=F32) -> torch.Tensor:
    """GEMM weight of the input-gradient convolution of the 3x3 convolution d[key] ([cout, 3, 3, cin] flattened):
    Wf[ci, (ky, kx), co] = W[co, (2 - ky, 2 - kx), ci]  (a derived copy in dt, cached with the others)."""
    k = key + (".flip" if dt == F32 else ".flip.bf16")
    if k not in d:
        d[k] = d[key].reshape(cout, 3, 3, cin).flip(1, 2).permute(3, 1, 2, 0).reshape(cin, 9 * cout).to(dt).contiguous()
    return d[k]


def _conv_wgrad(dy, x, *, n_img: int, R: int, cin: int, cout: int, x_img_stride: Optional[int] = None, x_off: int = 0,
                want_bias: bool = False, dyT=None):
    """dW [cout, 9*cin] (tap-major, ci fastest: the GEMM weight layout) = sum over output pixels of dy^T gathered(x), and db.
    dy [n_img*R*R, cout] plain rows; x is read as image i -> rows i*x_img_stride + x_off (a frame of the padded buffer).
    Returns (dW, db, dyT): the transposed dy can be reused for the other temporal taps."""
    dev, dt = dy.device, dy.dtype
    assert x.dtype == dt
    hw = R * R
    M = n_img * hw
    S, Mc = _split_plan(M, cout, 9 * cin)
    Mp = S * Mc
    if dyT is None:
        dyT = torch.empty(cout, Mp, device=dev, dtype=dt)
        ops.transpose(dy, dyT, M=M, Mp=Mp, C=cout, ldx=cout, ldy=Mp)
    xT = torch.empty(9 * cin, Mp, device=dev, dtype=dt)
    for t in range(9):
        ops.transpose(x, xT, M=M, Mp=Mp, C=cin, ldx=cin, ldy=Mp, y_row0=t * cin, out_h=R, out_w=R, in_h=R, in_w=R,
                      img_stride=hw if x_img_stride is None else x_img_stride, a_off=x_off, dy=t // 3 - 1, dx=t % 3 - 1)
    K = 9 * cin
    part = torch.empty(S, cout, K, device=dev, dtype=F32)
    ops.gemm(dyT, xT, part, M=cout, N=K, K=Mc, lda=Mp, ldy=K, ldw=Mp, n_split=S, a_split_stride=Mc, w_split_stride=Mc, y_split_stride=cout * K)
    dW = part[0] if S == 1 else ops.sum_partials(part, torch.empty(cout, K, device=dev, dtype=F32), stride=cout * K, n_part=S, n=cout * K)
    db = ops.row_sum(dyT, torch.empty(cout, device=dev, dtype=F32), ld=Mp, n=M, rows=cout) if want_bias else None
    return dW, db, dyT


def _w4(dW: torch.Tensor, cout: int, cin: int) -> torch.Tensor:
    """[cout, 9*cin] GEMM layout -> nn.Conv2d's [cout, cin, 3, 3]."""
    return dW.view(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous()


def _conv2d_bwd(d, key, dy, x, *, B: int, R: int, cin: int, cout: int, want_bias: bool, want_dx: bool = True, dx_acc=None):
    """Backward of y = conv3x3(x; d[key]) on [B*R*R, .] rows: (dW [cout,cin,3,3], db, dx)."""
    dW, db, _ = _conv_wgrad(dy, x, n_img=B, R=R, cin=cin, cout=cout, want_bias=want_bias)
    dx = None
    if want_dx:
        dx = torch.empty(B * R * R, cin, device=dy.device, dtype=F32)
        extra = dict(residual=dx_acc, ldr=cin) if dx_acc is not None else {}
        _conv(dy, _flip(d, key, cin, cout), dx, n_img=B, H=R, W=R, cin=cout, cout=cin, k=3, **extra)
    return _w4(dW, cout, cin), db, dx


# ----------------------------------------------------------------------------------------------------------------- forward
def rand_forward(model, batch, tok, ma, B: int, lat_rows=None, L: int = 0):
    """tok int64 [B, L, hw] (all frames; MAGE+: lat_rows [B*L*hw, 8] fp32 latents instead), ma [B*hw, C] fp32 (MA encoder output)
    -> (modulated ma, tape).  tape['kl_sum'] [B]."""
    d = model._derived.get(model._build)
    da = model.adain._derived.get(model.adain._build)
    R, Cc = model.image_resolution, model.vision_width
    hw, dev = R * R, ma.device
    blocks: list = []
    prior = model._video_prior(tok, lat_rows, B, L, tape=blocks, dt=model._dt())           # [B*hw, Cp] fp32
    Cp = prior.shape[1]
    mu = _conv(prior, d["mu2.w"], torch.empty(B * hw, 64, device=dev, dtype=F32), n_img=B, H=R, W=R, cin=Cp, cout=64, k=3, bias=d["mu2.b"])
    logvar = _conv(prior, d["var2.w"], torch.empty_like(mu), n_img=B, H=R, W=R, cin=Cp, cout=64, k=3, bias=d["var2.b"])
    eps = batch.get("reparam_noise")                                                        # [B,64,h,w]; else torch.randn (:571)
    if eps is None:
        eps = torch.randn(B, 64, R, R, device=dev)
    eps = eps.to(dev).float().permute(0, 2, 3, 1).reshape(B * hw, 64).contiguous()
    kl_sum = torch.empty(B, device=dev, dtype=F32)
    z = ops.reparam_kl(mu.view(B, -1), logvar.view(B, -1), eps.view(B, -1), torch.empty_like(mu).view(B, -1), kl_sum).view(B * hw, 64)
    y = _conv(z, d["conv_d2"], torch.empty(B * hw, Cc, device=dev, dtype=F32), n_img=B, H=R, W=R, cin=64, cout=Cc, k=3)
    g0 = _conv(y, da["mu0.w"], torch.empty(B * hw, Cc, device=dev, dtype=F32), n_img=B, H=R, W=R, cin=Cc, cout=Cc, k=3, bias=da["mu0.b"])
    gam = _conv(g0, da["mu1.w"], torch.empty_like(g0), n_img=B, H=R, W=R, cin=Cc, cout=Cc, k=3, bias=da["mu1.b"])
    b0 = _conv(y, da["var0.w"], torch.empty_like(g0), n_img=B, H=R, W=R, cin=Cc, cout=Cc, k=3, bias=da["var0.b"])
    bet = _conv(b0, da["var1.w"], torch.empty_like(g0), n_img=B, H=R, W=R, cin=Cc, cout=Cc, k=3, bias=da["var1.b"])
    out = ops.adain(ma, gam, bet, torch.empty_like(ma), B=B, P=hw, Cc=Cc, eps=model.adain.norm.eps)
    tape = dict(blocks=blocks, prior=prior, mu=mu, logvar=logvar, eps=eps, z=z, y=y, g0=g0, gam=gam, b0=b0, ma=ma, kl_sum=kl_sum, tok=tok,
                lat_rows=lat_rows, L=L if tok is None else tok.shape[1], B=B)
    return out, tape


# ----------------------------------------------------------------------------------------------------------------- backward
def _prior_backward(model, d, blocks, dprior, grads: Dict[str, torch.Tensor], B: int):
    """Backward of MAGE._video_prior: returns the gradient of block 0's padded input buffer (rows of the token embeddings)."""
    R = model.image_resolution
    hw, dev = R * R, dprior.device
    dout, (ds, dof) = dprior, blocks[-1]["out_map"]
    for i in reversed(range(len(blocks))):
        t, blk, pre = blocks[i], model.conv3d[i], f"conv3d.{i}"
        Lout, Lv, cin, cout = t["Lout"], t["Lv"], t["cin"], t["cout"]
        gn = dict(n_samples=B, rows_per_sample=Lout * hw, groups=16)
        # out = relu(GN2(c2) + res)
        dc2 = torch.zeros_like(t["c2"])
        dg, db, dres = ops.groupnorm_bwd(t["c2"], d[f"p{i}.g2.w"], d[f"p{i}.g2.b"], t["st2"], dout, dc2, sample_stride_rows=(Lout + 2) * hw,
                                         row_off=0, act=1, residual=t["res"], dy_sample_stride_rows=ds, dy_row_off=dof, want_dres=True, **gn)
        grads[pre + ".bn2.weight"], grads[pre + ".bn2.bias"] = dg, db
        del dout

        def conv3_bwd(name, dy, x, n_img, s_t, ci, co, want_dx=True):
            """Conv3d 3x3x3 (temporal stride s_t) as in MAGE._video_prior.conv3: weight gradient [co, ci, 3, 3, 3] and, accumulated
            over the temporal taps, the gradient (fp32) of the padded input buffer x.  The GEMM operands are in the tape's dtype."""
            dt = t["dt"]
            if dt != F32:
                dy = ops.cast(dy, torch.empty(dy.shape, device=dev, dtype=dt))
            dws, dyT = [], None
            for kd in range(3):
                dW, _, dyT = _conv_wgrad(dy, x, n_img=n_img, R=R, cin=ci, cout=co, x_img_stride=s_t * hw, x_off=kd * hw, dyT=dyT)
                dws.append(_w4(dW, co, ci))
            del dyT
            dx = None
            if want_dx:
                dx = torch.zeros(x.shape, device=dev, dtype=F32)
                for kd in range(3):
                    _conv(dy, _flip(d, f"p{i}.{name}.{kd}", ci, co, dt), dx, n_img=n_img, H=R, W=R, cin=co, cout=ci, k=3, y_img_stride=s_t * hw,
                          y_off=kd * hw, residual=dx, ldr=ci)
            return torch.stack(dws, 2), dx

        grads[pre + ".conv2.weight"], dxb = conv3_bwd("c2", dc2, t["xb"], B * (Lout + 2), 1, cout, cout)
        del dc2
        # xb = relu(GN1(c1)) in the stride-1 padded layout
        dc1 = torch.zeros_like(t["c1"])
        dg, db, _ = ops.groupnorm_bwd(t["c1"], d[f"p{i}.g1.w"], d[f"p{i}.g1.b"], t["st1"], dxb, dc1, sample_stride_rows=Lv * hw, row_off=0,
                                      act=1, dy_sample_stride_rows=(Lout + 2) * hw, dy_row_off=hw, **gn)
        grads[pre + ".bn1.weight"], grads[pre + ".bn1.bias"] = dg, db
        del dxb
        # res = GNd(cd)
        dcd = torch.zeros_like(t["cd"])
        dg, db, _ = ops.groupnorm_bwd(t["cd"], d[f"p{i}.gd.w"], d[f"p{i}.gd.b"], t["std"], dres, dcd, sample_stride_rows=Lv * hw, row_off=0,
                                      act=0, **gn)
        grads[pre + ".downsample.1.weight"], grads[pre + ".downsample.1.bias"] = dg, db
        del dres
        grads[pre + ".conv1.weight"], dxa1 = conv3_bwd("c1", dc1, t["xa"], B * Lv, 2, cin, cout)
        del dc1
        grads[pre + ".downsample.0.weight"], dxa2 = conv3_bwd("ds", dcd, t["xa"], B * Lv, 2, cin, cout)
        del dcd
        dxa1 += dxa2                                                        # the two branches of the block read the same buffer
        del dxa2
        dout, ds, dof = dxa1, 2 * Lv * hw, hw
        blocks[i] = None
    return dout, ds


def rand_backward(model, tape, dout_ma, kl_coef: torch.Tensor, grads: Dict[str, torch.Tensor], acc: Dict[str, Optional[torch.Tensor]]):
    """dout_ma = d loss / d (ADAIN output) [B*hw, C]; kl_coef = 1-element device tensor (d loss / d kl) / B.  Fills the gradients of
    conv3d.*, conv_mu2 / conv_var2, conv_d2, adain.*; adds the video prior's share to acc['emb'] (MAGE+: acc['emb_lin.*']); returns d loss / d (MA encoder output)."""
    d = model._derived.get(model._build)
    da = model.adain._derived.get(model.adain._build)
    R, Cc, B = model.image_resolution, model.vision_width, tape["B"]
    hw, dev = R * R, dout_ma.device
    dma, dgam = ops.adain_bwd(tape["ma"], tape["gam"], dout_ma, B=B, P=hw, Cc=Cc, eps=model.adain.norm.eps)
    dbet = dout_ma
    geo = dict(B=B, R=R, cin=Cc, cout=Cc, want_bias=True)
    gw, gb, dg0 = _conv2d_bwd(da, "mu1.w", dgam, tape["g0"], **geo)
    grads["adain.conv_mu.1.weight"], grads["adain.conv_mu.1.bias"] = gw, gb
    gw, gb, dy = _conv2d_bwd(da, "mu0.w", dg0, tape["y"], **geo)
    grads["adain.conv_mu.0.weight"], grads["adain.conv_mu.0.bias"] = gw, gb
    gw, gb, db0 = _conv2d_bwd(da, "var1.w", dbet, tape["b0"], **geo)
    grads["adain.conv_var.1.weight"], grads["adain.conv_var.1.bias"] = gw, gb
    gw, gb, dy = _conv2d_bwd(da, "var0.w", db0, tape["y"], dx_acc=dy, **geo)
    grads["adain.conv_var.0.weight"], grads["adain.conv_var.0.bias"] = gw, gb
    del dgam, dg0, db0
    gw, _, dz = _conv2d_bwd(d, "conv_d2", dy, tape["z"], B=B, R=R, cin=64, cout=Cc, want_bias=False)
    grads["conv_d2.weight"] = gw
    dmu, dlv = ops.reparam_kl_bwd(tape["mu"], tape["logvar"], tape["eps"], dz, kl_coef)
    Cp = tape["prior"].shape[1]
    gw, gb, dprior = _conv2d_bwd(d, "mu2.w", dmu, tape["prior"], B=B, R=R, cin=Cp, cout=64, want_bias=True)
    grads["conv_mu2.weight"], grads["conv_mu2.bias"] = gw, gb
    gw, gb, dprior = _conv2d_bwd(d, "var2.w", dlv, tape["prior"], B=B, R=R, cin=Cp, cout=64, want_bias=True, dx_acc=dprior)
    grads["conv_var2.weight"], grads["conv_var2.bias"] = gw, gb
    dxa, ds = _prior_backward(model, d, tape["blocks"], dprior, grads, B)
    tok, L = tape["tok"], tape["L"]
    if tok is not None:
        if acc.get("emb") is None:
            acc["emb"] = torch.zeros(model.codebook_size, Cc, device=dev, dtype=F32)
        ops.embedding_bwd(tok.reshape(-1).contiguous(), dxa, acc["emb"], group=L * hw, group_stride=ds, off=hw)
    else:                                                   # MAGE+: the embeddings are Linear(latents), written into the same frame slots
        lat = tape["lat_rows"]
        dW, db = _wgrad(dxa, lat, M=B * L * hw, N=Cc, K=lat.shape[1], ld_dy=Cc, ld_x=lat.shape[1],
                        dy_geo=dict(out_w=L * hw, img_stride=ds, a_off=hw))
        acc["emb_lin.w"] = dW if acc.get("emb_lin.w") is None else acc["emb_lin.w"] + dW
        acc["emb_lin.b"] = db if acc.get("emb_lin.b") is None else acc["emb_lin.b"] + db
    return dma
