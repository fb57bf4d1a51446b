"""Stage-1 training of the f4 VQ-VAE on the HIP kernels (reference train_vqvae.py:13-35).

``x_tilde, z_e_x, z_q_x = model(images)`` in training mode returns tensors that carry ONE autograd node
(``VQVAEForwardFn``): the caller's loss -- ``mse(x_tilde, x) + mse(z_q_x, z_e_x.detach()) + beta * mse(z_e_x, z_q_x.detach())`` --
back-propagates into it, and its ``backward`` produces every parameter gradient with libmage_hip.so:

* BatchNorm2d on BATCH statistics (column reductions over channels-last rows, two-pass variance) and its backward;
* the straight-through estimator of ``VectorQuantizationStraightThrough`` (vqvae_model.py:34-65): the decoder's input gradient
  passes to z_e unchanged, the codebook receives ``index_add`` of the gradient of z_q_x (a scatter kernel);
* every convolution gradient as MFMA GEMMs: data gradients are the forward implicit-GEMM kernel on re-laid weights (a stride-2
  convolution's is four sub-pixel GEMMs, a transposed convolution's is a stride-2 gather GEMM), weight gradients are gathered
  transposes + one split-K launch (mage_train._wgrad machinery);
* the in-place-ReLU quirk of ResBlock (out = relu(x) + f(relu(x)), vqvae_model.py:111-124) exactly as in inference.

fp32 throughout (stage-1 training is small: 2 x 2.4 M parameters).

The f8 (CATER) stack (vqvae_model.py:126-165,192-214; ``train_vqvae.py --dataset cater-gen``) has no BatchNorm: its training pass is
the inference kernels with the activations kept, and its backward adds the gradients of the bottleneck blocks (1x1 / 3x3 convolutions
behind out-of-place ReLUs, the 1x1 identity path), of MaxPool2d(2) (first maximum of the window, PyTorch's tie rule) and nearest
Upsample (2x2 block sums), of the 7x7 three-channel stem (weight gradient over the image padded to 8 channels) and of the 1x1
three-channel tanh head.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import nn

from .. import ops
from .mage_train import _split_plan

F32 = torch.float32

__all__ = ["VQVAEForwardFn", "vq_train_forward", "vq_train_backward", "vq8_train_forward", "vq8_train_backward"]


# ----------------------------------------------------------------------------------------------------------------- helpers
def _conv(a, wgt, y, *, n_img, H, W, cin, cout, k, stride=1, pad=None, OH=None, OW=None, **epi):
    pad = k // 2 if pad is None else pad
    OH = H if OH is None else OH
    OW = W if OW is None else OW
    return ops.gemm(a, wgt, y, M=n_img * OH * OW, N=cout, K=k * k * cin, lda=cin, ldy=cout, out_h=OH, out_w=OW, in_h=H, in_w=W,
                    taps_h=k, taps_w=k, cin=cin, stride=stride, dy0=-pad, dx0=-pad, **epi)


def _wgrad_taps(dy, x, *, M: int, N: int, Cin: int, taps, grid, dy_geo=None, x_ld=None):
    """dW[N, ntaps*Cin] = sum_m dy[m, :]^T x[gather_tap(m), :] and db[N]: `taps` = list of dict(dy, dx) gathers of x over the output
    grid `grid` = dict(out_h, out_w, in_h, in_w, stride) (x is [n_img*in_h*in_w, Cin] channels-last); dy_geo optionally gathers the
    rows of dy itself (the sub-pixel positions of a transposed convolution's output)."""
    dev = dy.device
    nt = len(taps)
    S, Mc = _split_plan(M, N, nt * Cin)
    Mp = S * Mc
    dyT = torch.empty(N, Mp, device=dev, dtype=F32)
    ops.transpose(dy, dyT, M=M, Mp=Mp, C=N, ldx=N, ldy=Mp, **(dy_geo or {}))
    xT = torch.empty(nt * Cin, Mp, device=dev, dtype=F32)
    for i, t in enumerate(taps):
        ops.transpose(x, xT, M=M, Mp=Mp, C=Cin, ldx=Cin if x_ld is None else x_ld, ldy=Mp, y_row0=i * Cin, out_h=grid["out_h"],
                      out_w=grid["out_w"], in_h=grid["in_h"], in_w=grid["in_w"], img_stride=grid["in_h"] * grid["in_w"], dy=t["dy"], dx=t["dx"],
                      stride=grid.get("stride", 1))
    K = nt * Cin
    part = torch.empty(S, N, K, device=dev, dtype=F32)
    ops.gemm(dyT, xT, part, M=N, N=K, K=Mc, lda=Mp, ldy=K, ldw=Mp, n_split=S, a_split_stride=Mc, w_split_stride=Mc, y_split_stride=N * K)
    dW = part[0] if S == 1 else ops.sum_partials(part, torch.empty(N, K, device=dev, dtype=F32), stride=N * K, n_part=S, n=N * K)
    db = ops.row_sum(dyT, torch.empty(N, device=dev, dtype=F32), ld=Mp, n=M, rows=N)
    return dW, db


def _bn_forward(bn: nn.BatchNorm2d, x_rows, relu: bool, residual=None):
    """Training-mode BatchNorm2d on channels-last rows: batch statistics, running-statistics update (momentum, unbiased variance),
    y = [relu](norm(x) * gamma + beta [+ residual]).  Returns (y, saved)."""
    mean, var, rstd = ops.bn_train_stats(x_rows, bn.eps)
    rows = x_rows.shape[0]
    with torch.no_grad():                                          # [C]-sized buffer bookkeeping (nn.BatchNorm2d semantics)
        if bn.track_running_stats and bn.running_mean is not None:
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(var * (rows / max(rows - 1, 1)), alpha=mom)
    g, b = bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous()
    y = ops.bn_apply(x_rows, mean, rstd, g, b, torch.empty_like(x_rows), relu, residual=residual)
    return y, dict(x=x_rows, mean=mean, rstd=rstd, gamma=g)


def _res_forward(vq, w, p: str, rb, r, n_img: int, H: int, W: int, post_relu: bool):
    """ResBlock on an already-ReLU'd input r (in-place-ReLU quirk): out = [relu](r + BN(conv1(relu(BN(conv3(r))))))."""
    D = vq.dim
    c3 = _conv(r, w[p + ".w3.f32"], torch.empty_like(r), n_img=n_img, H=H, W=W, cin=D, cout=D, k=3, bias=w[p + ".b3"])
    t, s3 = _bn_forward(rb.block[2], c3, relu=True)
    c1 = _conv(t, w[p + ".w1.f32"], torch.empty_like(r), n_img=n_img, H=H, W=W, cin=D, cout=D, k=1, bias=w[p + ".b1"])
    out, s1 = _bn_forward(rb.block[5], c1, relu=post_relu, residual=r)
    return out, dict(r=r, t=t, s3=s3, s1=s1, out=out, post_relu=post_relu)


_TAPS3 = [dict(dy=ky - 1, dx=kx - 1) for ky in range(3) for kx in range(3)]


def _res_backward(vq, w, p: str, rb, key: str, tape, dout, n_img: int, H: int, W: int, grads: Dict[str, torch.Tensor]):
    """Returns d/dr (r = the block's ReLU'd input); fills the block's parameter gradients (keys `key`.block.{1,2,4,5}.*)."""
    D, dev = vq.dim, dout.device
    M = n_img * H * W
    do = ops.act_bwd(tape["out"], dout, torch.empty_like(dout), ops.ACT_RELU) if tape["post_relu"] else dout
    s1, s3 = tape["s1"], tape["s3"]
    dc1 = torch.empty_like(do)
    grads[f"{key}.block.5.weight"], grads[f"{key}.block.5.bias"] = ops.bn_backward(s1["x"], do, s1["mean"], s1["rstd"], s1["gamma"], dc1)
    dW1, db1 = _wgrad_taps(dc1, tape["t"], M=M, N=D, Cin=D, taps=[dict(dy=0, dx=0)], grid=dict(out_h=H, out_w=W, in_h=H, in_w=W))
    grads[f"{key}.block.4.weight"], grads[f"{key}.block.4.bias"] = dW1.view(D, D, 1, 1), db1
    w1T = w[p + ".w1.f32"].t().contiguous()                                     # [Cin, Cout]: dX = dY W
    dt = ops.gemm(dc1, w1T, torch.empty_like(do), M=M, N=D, K=D, lda=D, ldy=D)
    dc3 = torch.empty_like(do)
    grads[f"{key}.block.2.weight"], grads[f"{key}.block.2.bias"] = ops.bn_backward(s3["x"], dt, s3["mean"], s3["rstd"], s3["gamma"], dc3,
                                                                                  mask=tape["t"])
    dW3, db3 = _wgrad_taps(dc3, tape["r"], M=M, N=D, Cin=D, taps=_TAPS3, grid=dict(out_h=H, out_w=W, in_h=H, in_w=W))
    grads[f"{key}.block.1.weight"] = dW3.view(D, 3, 3, D).permute(0, 3, 1, 2).contiguous()       # [Cout, Cin, kh, kw]
    grads[f"{key}.block.1.bias"] = db3
    w3f = w[p + ".w3.f32"].view(D, 3, 3, D).flip(1, 2).permute(3, 1, 2, 0).reshape(D, 9 * D).contiguous()   # input-gradient conv
    return _conv(dc3, w3f, torch.empty_like(do), n_img=n_img, H=H, W=W, cin=D, cout=D, k=3, residual=do, ldr=D)


def _subpixel_weights(wt: torch.Tensor):
    """ConvTranspose2d(cin, cout, 4, 2, 1) weight [cin, cout, 4, 4] -> four 2x2 sub-pixel convolution weights [cout, (k2y,k2x,ci)] and
    the (ky, kx) each (py, px, k2y, k2x) came from (the construction of VectorQuantizedVAE._build for decoder[3])."""
    out = {}
    for py in range(2):
        for px in range(2):
            kys = [py + 1 - 2 * (py - k2) for k2 in range(2)]
            kxs = [px + 1 - 2 * (px - k2) for k2 in range(2)]
            sub = wt[:, :, kys][:, :, :, kxs]
            out[(py, px)] = (sub.permute(1, 2, 3, 0).reshape(wt.shape[1], -1).contiguous(), kys, kxs)
    return out


def _convt_forward(x_rows, subw, bias, n_img: int, h: int, wd: int, cin: int, cout: int):
    """ConvTranspose2d(cin, cout, 4, 2, 1) as four sub-pixel GEMMs: [n*h*wd, cin] -> raw [n*4*h*wd, cout] (+ bias)."""
    up = torch.empty(n_img * 4 * h * wd, cout, device=x_rows.device, dtype=F32)
    for (py, px), (ws, _, _) in subw.items():
        ops.gemm(x_rows, ws, up, M=n_img * h * wd, N=cout, K=4 * cin, lda=cin, ldy=cout, out_h=h, out_w=wd, in_h=h, in_w=wd, taps_h=2,
                 taps_w=2, cin=cin, stride=1, dy0=py, dx0=px, dys=-1, dxs=-1, y_img_stride=4 * h * wd, y_mul_y=4 * wd, y_mul_x=2,
                 y_off=py * 2 * wd + px, bias=bias)
    return up


# ----------------------------------------------------------------------------------------------------------------- forward
def vq_train_forward(vq, x: torch.Tensor):
    """VectorQuantizedVAE.forward in training mode (vqvae_model.py:244-248): (x_tilde NCHW, z_e rows, z_q rows, tape)."""
    assert vq.down_ratio == 4
    w = vq._weights()
    x = x.float().contiguous()
    N, Cin, H, W = x.shape
    if H % 4 or W % 4:
        raise ValueError(f"f4 VQ-VAE input {H}x{W} must be a multiple of 4 in both dimensions")
    D, dev = vq.dim, x.device
    h, wd = H // 4, W // 4
    enc, dec = vq.encoder, vq.decoder
    a0 = ops.conv_in(x, w["e0.wt"], w["e0.b"], None, None, torch.empty(N * (H // 2) * (W // 2), D, device=dev, dtype=F32), cin=Cin, H=H, W=W,
                     cout=D, kh=4, kw=4, stride=2, pad=1)
    h0, s0 = _bn_forward(enc[1], a0, relu=True)
    r1 = _conv(h0, w["e3.w.f32"], torch.empty(N * h * wd, D, device=dev, dtype=F32), n_img=N, H=H // 2, W=W // 2, cin=D, cout=D, k=4, stride=2,
               pad=1, OH=h, OW=wd, bias=w["e3.b"], act=ops.ACT_RELU)                     # conv + the ResBlock's in-place ReLU
    r2, t4 = _res_forward(vq, w, "e4", enc[4], r1, N, h, wd, post_relu=True)
    z_e, t5 = _res_forward(vq, w, "e5", enc[5], r2, N, h, wd, post_relu=False)
    ids = ops.vq_nearest(z_e, w["cbt"], w["c2"])
    zq = ops.embedding(ids, w["cb"], torch.empty(N * h * wd, D, device=dev, dtype=F32))
    rd = ops.relu(zq, torch.empty_like(zq))                                              # decoder[0]'s in-place ReLU on z_q_x_st
    rd1, td0 = _res_forward(vq, w, "d0", dec[0], rd, N, h, wd, post_relu=True)
    u, td1 = _res_forward(vq, w, "d1", dec[1], rd1, N, h, wd, post_relu=True)            # decoder[2] ReLU folded
    subw = _subpixel_weights(dec[3].weight.detach().float())
    ct = _convt_forward(u, subw, w["d3.b"], N, h, wd, D, D)
    v, s4 = _bn_forward(dec[4], ct, relu=True)
    nt = 16 * Cin
    taps = ops.gemm(v, w["d6.w16.f32"], torch.empty(N * 4 * h * wd, nt, device=dev, dtype=F32), M=N * 4 * h * wd, N=nt, K=D, lda=D, ldy=nt)
    x_tilde = ops.convt_fold_tanh(taps, w["d6.b"], torch.empty(N, Cin, H, W, device=dev, dtype=F32), N=N, IH=2 * h, IW=2 * wd, cout=Cin)
    tape = dict(x=x, a0s=s0, h0=h0, r1=r1, t4=t4, t5=t5, ids=ids, rd=rd, td0=td0, td1=td1, u=u, subw=subw, s4=s4, v=v, x_tilde=x_tilde,
                N=N, H=H, W=W, h=h, wd=wd, Cin=Cin)
    return x_tilde, z_e, zq, tape


# ----------------------------------------------------------------------------------------------------------------- backward
def vq_train_backward(vq, tape, g_xt, g_ze_rows, g_zq_rows) -> Dict[str, torch.Tensor]:
    """Parameter gradients (state_dict names) from the gradients of the three outputs (any may be None)."""
    w = vq._weights()
    N, H, W, h, wd, Cin, D = tape["N"], tape["H"], tape["W"], tape["h"], tape["wd"], tape["Cin"], vq.dim
    dev = tape["x"].device
    enc, dec = vq.encoder, vq.decoder
    grads: Dict[str, torch.Tensor] = {}
    M = N * h * wd
    dz_e = g_ze_rows.clone() if g_ze_rows is not None else torch.zeros(M, D, device=dev, dtype=F32)
    if g_zq_rows is not None:                                      # z_q_x_bar: index_add into the codebook (vqvae_model.py:54-62)
        gcb = torch.zeros(vq.K, D, device=dev, dtype=F32)
        ops.embedding_bwd(tape["ids"].reshape(-1), g_zq_rows, gcb)
        grads["codebook.embedding.weight"] = gcb
    if g_xt is not None:
        M4, nt = N * 4 * h * wd, 16 * Cin
        g_xt = g_xt.float().contiguous()
        ds = ops.act_bwd(tape["x_tilde"], g_xt, torch.empty_like(g_xt), ops.ACT_TANH)               # d tanh from its output
        grads["decoder.6.bias"] = ops.row_sum(ds, torch.empty(N * Cin, device=dev, dtype=F32), ld=H * W, n=H * W,
                                              rows=N * Cin).view(N, Cin).sum(0)                    # [N, Cin] -> [Cin] (tiny)
        dtaps = ops.convt_unfold_tanh_bwd(ds, None, torch.empty(M4, nt, device=dev, dtype=F32), N=N, IH=2 * h, IW=2 * wd, cout=Cin)
        dW16, _ = _wgrad_taps(dtaps, tape["v"], M=M4, N=nt, Cin=D, taps=[dict(dy=0, dx=0)], grid=dict(out_h=2 * h, out_w=2 * wd, in_h=2 * h,
                                                                                                  in_w=2 * wd))
        grads["decoder.6.weight"] = dW16.view(4, 4, Cin, D).permute(3, 2, 0, 1).contiguous()       # [cin, cout, ky, kx]
        w16T = w["d6.w16.f32"].t().contiguous()                                                    # [D, 16*Cin]
        dv = ops.gemm(dtaps, w16T, torch.empty(M4, D, device=dev, dtype=F32), M=M4, N=D, K=nt, lda=nt, ldy=D)
        s4 = tape["s4"]
        dct = torch.empty(M4, D, device=dev, dtype=F32)
        grads["decoder.4.weight"], grads["decoder.4.bias"] = ops.bn_backward(s4["x"], dv, s4["mean"], s4["rstd"], s4["gamma"], dct, mask=tape["v"])
        # ConvTranspose2d decoder[3]: bias, weight (per sub-pixel), input gradient (stride-2 gather GEMMs over dct)
        grads["decoder.3.bias"] = ops.group_rowsum(dct, torch.empty(1, D, device=dev, dtype=F32), rows=M4, C=D, div=1, mod=1).view(D)
        dwt = torch.zeros(D, D, 4, 4, device=dev, dtype=F32)                                       # [cin, cout, ky, kx]
        du = None
        for (py, px), (ws, kys, kxs) in tape["subw"].items():
            sub_rows = dict(out_h=h, out_w=wd, in_h=2 * h, in_w=2 * wd, img_stride=4 * h * wd, dy=py, dx=px, stride=2)   # dct at (2oy+py, 2ox+px)
            taps = [dict(dy=py - k2y, dx=px - k2x) for k2y in range(2) for k2x in range(2)]
            dWs, _ = _wgrad_taps(dct, tape["u"], M=M, N=D, Cin=D, taps=taps, grid=dict(out_h=h, out_w=wd, in_h=h, in_w=wd), dy_geo=sub_rows)
            dWs = dWs.view(D, 2, 2, D)                                                             # [co, k2y, k2x, ci]
            for k2y in range(2):
                for k2x in range(2):
                    dwt[:, :, kys[k2y], kxs[k2x]] = dWs[:, k2y, k2x, :].t()                        # layout plumbing back to [ci, co, ky, kx]
            wsT = ws.view(D, 2, 2, D).permute(3, 1, 2, 0).reshape(D, 4 * D).contiguous()           # [ci, (k2y, k2x, co)]
            du_new = torch.empty(M, D, device=dev, dtype=F32)
            ops.gemm(dct, wsT, du_new, M=M, N=D, K=4 * D, lda=D, ldy=D, out_h=h, out_w=wd, in_h=2 * h, in_w=2 * wd, taps_h=2, taps_w=2, cin=D,
                     stride=2, dy0=-py, dx0=-px, dys=2, dxs=2, a_img_stride=4 * h * wd, residual=du, ldr=D)
            du = du_new
        grads["decoder.3.weight"] = dwt
        drd1 = _res_backward(vq, w, "d1", dec[1], "decoder.1", tape["td1"], du, N, h, wd, grads)
        drd = _res_backward(vq, w, "d0", dec[0], "decoder.0", tape["td0"], drd1, N, h, wd, grads)
        dzq_st = ops.act_bwd(tape["rd"], drd, torch.empty_like(drd), ops.ACT_RELU)                  # through decoder[0]'s in-place ReLU
        if g_ze_rows is not None:                                                                   # straight-through (vqvae_model.py:47-49)
            ops.dropout(dzq_st, dz_e, 0.0, 0, accumulate=True)                                      # dz_e += dzq_st (p = 0: a plain add)
        else:
            dz_e = dzq_st
    # ---- encoder
    dr2 = _res_backward(vq, w, "e5", enc[5], "encoder.5", tape["t5"], dz_e, N, h, wd, grads)
    dr1 = _res_backward(vq, w, "e4", enc[4], "encoder.4", tape["t4"], dr2, N, h, wd, grads)
    da3 = ops.act_bwd(tape["r1"], dr1, torch.empty_like(dr1), ops.ACT_RELU)
    H2, W2 = H // 2, W // 2
    taps16 = [dict(dy=ky - 1, dx=kx - 1) for ky in range(4) for kx in range(4)]
    dW3, db3 = _wgrad_taps(da3, tape["h0"], M=M, N=D, Cin=D, taps=taps16, grid=dict(out_h=h, out_w=wd, in_h=H2, in_w=W2, stride=2))
    grads["encoder.3.weight"] = dW3.view(D, 4, 4, D).permute(0, 3, 1, 2).contiguous()
    grads["encoder.3.bias"] = db3
    # input gradient of the stride-2 convolution = the transposed convolution with the same weight tensor read as [cin_T = co, cout_T = ci]
    sub_e3 = _subpixel_weights(enc[3].weight.detach().float())
    dh0 = _convt_forward(da3, sub_e3, None, N, h, wd, D, D)
    s0 = tape["a0s"]
    da0 = torch.empty_like(dh0)
    grads["encoder.1.weight"], grads["encoder.1.bias"] = ops.bn_backward(s0["x"], dh0, s0["mean"], s0["rstd"], s0["gamma"], da0, mask=tape["h0"])
    # stem conv (cin = Cin <= 4): the image as channels-last rows is NCHW itself when Cin == 1; more channels: rows padded to 8 (plumbing)
    M0 = N * H2 * W2
    if Cin == 1:
        xr, Cp = tape["x"].view(N * H * W, 1), 1
    else:
        Cp = 8
        xr = torch.zeros(N * H * W, Cp, device=dev, dtype=F32)
        xr[:, :Cin] = tape["x"].permute(0, 2, 3, 1).reshape(N * H * W, Cin)
    dW0, db0 = _wgrad_taps(da0, xr, M=M0, N=D, Cin=Cp, taps=taps16, grid=dict(out_h=H2, out_w=W2, in_h=H, in_w=W, stride=2), x_ld=Cp)
    grads["encoder.0.weight"] = dW0.view(D, 4, 4, Cp)[..., :Cin].permute(0, 3, 1, 2).contiguous()
    grads["encoder.0.bias"] = db0
    return grads


# ================================================================================================================= f8 (CATER) stack
def _kconv_bwd(w, wkey: str, dy, x, *, n_img: int, H: int, W: int, cin: int, cout: int, k: int, want_dx: bool = True, dx_res=None):
    """Backward of y = Conv2d(cin, cout, k, 1, k // 2)(x) on channels-last rows: (dW [cout, cin, k, k], db, dx [+ dx_res])."""
    M, r = n_img * H * W, k // 2
    taps = [dict(dy=ky - r, dx=kx - r) for ky in range(k) for kx in range(k)]
    dW, db = _wgrad_taps(dy, x, M=M, N=cout, Cin=cin, taps=taps, grid=dict(out_h=H, out_w=W, in_h=H, in_w=W))
    dW = dW.view(cout, k, k, cin).permute(0, 3, 1, 2).contiguous()
    dx = None
    if want_dx:
        dx = torch.empty(M, cin, device=dy.device, dtype=F32)
        extra = dict(residual=dx_res, ldr=cin) if dx_res is not None else {}
        if k == 1:
            ops.gemm(dy, w[wkey].t().contiguous(), dx, M=M, N=cin, K=cout, lda=cout, ldy=cin, **extra)
        else:
            wf = w[wkey].view(cout, k, k, cin).flip(1, 2).permute(3, 1, 2, 0).reshape(cin, k * k * cout).contiguous()
            _conv(dy, wf, dx, n_img=n_img, H=H, W=W, cin=cout, cout=cin, k=k, **extra)
    return dW, db, dx


def _bott_forward(vq, w, p: str, x, n_img: int, H: int, W: int, cin: int, cout: int, ks, post_relu: bool):
    """EncoderBlock / DecoderBlock (vqvae_model.py:126-165): id_path(x) + conv(relu(conv(relu(conv(relu(conv(relu(x)))))))), the ReLUs
    out of place (the identity path sees x itself).  ks = the four kernel sizes.  Returns (out, tape)."""
    hid, dev = cout // 4, x.device
    M = n_img * H * W
    xr = ops.relu(x, torch.empty_like(x))
    chans = [cin, hid, hid, hid, cout]
    hs = [xr]
    for j in range(3):
        hs.append(_conv(hs[-1], w[f"{p}.w{2 * j + 1}.f32"], torch.empty(M, chans[j + 1], device=dev, dtype=F32), n_img=n_img, H=H, W=W,
                        cin=chans[j], cout=chans[j + 1], k=ks[j], bias=w[f"{p}.b{2 * j + 1}"], act=ops.ACT_RELU))
    has_id = (p + ".wid.f32") in w
    idp = _conv(x, w[p + ".wid.f32"], torch.empty(M, cout, device=dev, dtype=F32), n_img=n_img, H=H, W=W, cin=cin, cout=cout, k=1,
                bias=w[p + ".bid"]) if has_id else x
    out = _conv(hs[3], w[p + ".w7.f32"], torch.empty(M, cout, device=dev, dtype=F32), n_img=n_img, H=H, W=W, cin=hid, cout=cout, k=ks[3],
                bias=w[p + ".b7"], residual=idp, ldr=cout, post_relu=post_relu)
    return out, dict(x=x, hs=hs, out=out if post_relu else None, has_id=has_id, geo=(n_img, H, W, cin, cout), ks=ks)


def _bott_backward(vq, w, p: str, key: str, t, dout, grads: Dict[str, torch.Tensor]):
    """d/dx of the block; fills `key`.block.{1,3,5,7}.* and `key`.id_path.*."""
    n_img, H, W, cin, cout = t["geo"]
    hid, ks, hs = cout // 4, t["ks"], t["hs"]
    geo = dict(n_img=n_img, H=H, W=W)
    do = ops.act_bwd(t["out"], dout, torch.empty_like(dout), ops.ACT_RELU) if t["out"] is not None else dout
    chans = [cin, hid, hid, hid, cout]
    g = do
    for j in (3, 2, 1, 0):
        dW, db, dh = _kconv_bwd(w, f"{p}.w{2 * j + 1}.f32", g, hs[j], cin=chans[j], cout=chans[j + 1], k=ks[j], **geo)
        grads[f"{key}.block.{2 * j + 1}.weight"], grads[f"{key}.block.{2 * j + 1}.bias"] = dW, db
        g = ops.act_bwd(hs[j], dh, dh, ops.ACT_RELU)                       # hs[j] = relu(.): its sign is the mask (hs[0] = relu(x))
    if t["has_id"]:
        dW, db, dx = _kconv_bwd(w, p + ".wid.f32", do, t["x"], cin=cin, cout=cout, k=1, dx_res=g, **geo)
        grads[f"{key}.id_path.weight"], grads[f"{key}.id_path.bias"] = dW, db
        return dx
    return ops.dropout(do, g, 0.0, 0, accumulate=True)                     # g += do (p = 0: a plain add)


_F8_ENC = ((1, 1, 1), (3, 1, 1), (5, 1, 2), (7, 2, 4))                      # (module index, cin / dim, cout / dim)
_F8_DEC = ((0, 4, 2), (2, 2, 1), (4, 1, 1), (6, 1, 1))


def vq8_train_forward(vq, x: torch.Tensor):
    """VectorQuantizedVAE.forward of the f8 stack with the activations kept: (x_tilde NCHW, z_e rows, z_q rows, tape)."""
    w = vq._weights()
    x = x.float().contiguous()
    N, Cin, H, W = x.shape
    if H % 8 or W % 8:
        raise ValueError(f"f8 VQ-VAE input {H}x{W} must be a multiple of 8 in both dimensions")
    dim, dev = vq.dim, x.device
    h = vq._stem7(w, x)
    enc_t, pools = [], []
    hh, ww = H, W
    for bi, ci, co in _F8_ENC:
        last = bi == 7
        h, t = _bott_forward(vq, w, f"e{bi}", h, N, hh, ww, ci * dim, co * dim, (3, 3, 3, 1), post_relu=last)      # trailing nn.ReLU (:201)
        enc_t.append(t)
        if not last:
            pools.append((h, hh, ww, co * dim))
            h = ops.maxpool2(h, torch.empty(N * (hh // 2) * (ww // 2), co * dim, device=dev, dtype=F32), N=N, H=hh, W=ww, Cc=co * dim)
            hh, ww = hh // 2, ww // 2
    z_e, D = h, 4 * dim
    ids = ops.vq_nearest(z_e, w["cbt"], w["c2"])
    zq = ops.embedding(ids, w["cb"], torch.empty(N * hh * ww, D, device=dev, dtype=F32))
    dec_t, ups = [], []
    h = zq
    for bi, ci, co in _F8_DEC:
        last = bi == 6
        h, t = _bott_forward(vq, w, f"d{bi}", h, N, hh, ww, ci * dim, co * dim, (1, 3, 3, 3), post_relu=last)      # decoder[7] ReLU folded
        dec_t.append(t)
        if not last:
            ups.append((hh, ww, co * dim))
            h = ops.upsample2(h, torch.empty(N * 4 * hh * ww, co * dim, device=dev, dtype=F32), N=N, H=hh, W=ww, Cc=co * dim)
            hh, ww = hh * 2, ww * 2
    x_tilde = ops.conv_out(h, w["d8.wt"], w["d8.b"], torch.empty(N, Cin, H, W, device=dev, dtype=F32), N=N, IH=H, IW=W, cin=dim, cout=Cin,
                           transposed=False)
    tape = dict(f8=True, x=x, enc=enc_t, pools=pools, ids=ids, dec=dec_t, ups=ups, last=h, x_tilde=x_tilde, N=N, H=H, W=W, h=H // 8, wd=W // 8,
                Cin=Cin)
    return x_tilde, z_e, zq, tape


def vq8_train_backward(vq, tape, g_xt, g_ze_rows, g_zq_rows) -> Dict[str, torch.Tensor]:
    w = vq._weights()
    N, H, W, Cin, dim = tape["N"], tape["H"], tape["W"], tape["Cin"], vq.dim
    D, dev = 4 * dim, tape["x"].device
    grads: Dict[str, torch.Tensor] = {}
    M8 = N * tape["h"] * tape["wd"]
    dz_e = g_ze_rows.clone() if g_ze_rows is not None else torch.zeros(M8, D, device=dev, dtype=F32)
    if g_zq_rows is not None:                                      # z_q_x_bar: index_add into the codebook (vqvae_model.py:54-62)
        gcb = torch.zeros(vq.K, D, device=dev, dtype=F32)
        ops.embedding_bwd(tape["ids"].reshape(-1), g_zq_rows, gcb)
        grads["codebook.embedding.weight"] = gcb
    if g_xt is not None:
        M = N * H * W
        g_xt = g_xt.float().contiguous()
        ds = ops.act_bwd(tape["x_tilde"], g_xt, torch.empty_like(g_xt), ops.ACT_TANH)               # d tanh from its output, NCHW
        grads["decoder.8.bias"] = ops.row_sum(ds, torch.empty(N * Cin, device=dev, dtype=F32), ld=H * W, n=H * W,
                                              rows=N * Cin).view(N, Cin).sum(0)
        dsr = torch.zeros(M, 8, device=dev, dtype=F32)                                              # channels-last rows, padded to 8 (plumbing)
        dsr[:, :Cin] = ds.permute(0, 2, 3, 1).reshape(M, Cin)
        dW8, _ = _wgrad_taps(dsr, tape["last"], M=M, N=8, Cin=dim, taps=[dict(dy=0, dx=0)], grid=dict(out_h=H, out_w=W, in_h=H, in_w=W))
        grads["decoder.8.weight"] = dW8[:Cin].reshape(Cin, dim, 1, 1).contiguous()
        w8 = torch.zeros(8, dim, device=dev, dtype=F32)
        w8[:Cin] = w["d8.wt"]
        g = ops.gemm(dsr, w8.t().contiguous(), torch.empty(M, dim, device=dev, dtype=F32), M=M, N=dim, K=8, lda=8, ldy=dim)
        for (bi, ci, co), t in zip(reversed(_F8_DEC), reversed(tape["dec"])):
            if bi != 6:
                hh, ww, cc = tape["ups"].pop()
                g = ops.upsample2_bwd(g, N=N, H=hh, W=ww, Cc=cc)
            g = _bott_backward(vq, w, f"d{bi}", f"decoder.{bi}", t, g, grads)
        if g_ze_rows is not None:                                                                   # straight-through (vqvae_model.py:47-49)
            ops.dropout(g, dz_e, 0.0, 0, accumulate=True)
        else:
            dz_e = g
    g = dz_e
    for (bi, ci, co), t in zip(reversed(_F8_ENC), reversed(tape["enc"])):
        if bi != 7:
            xin, hh, ww, cc = tape["pools"].pop()
            g = ops.maxpool2_bwd(xin, g, N=N, H=hh, W=ww, Cc=cc)
        g = _bott_backward(vq, w, f"e{bi}", f"encoder.{bi}", t, g, grads)
    # 7x7 stem over the Cin-channel image: weight gradient on the image as channels-last rows padded to 8 channels (layout plumbing)
    M = N * H * W
    xr = torch.zeros(M, 8, device=dev, dtype=F32)
    xr[:, :Cin] = tape["x"].permute(0, 2, 3, 1).reshape(M, Cin)
    taps = [dict(dy=ky - 3, dx=kx - 3) for ky in range(7) for kx in range(7)]
    dW0, db0 = _wgrad_taps(g, xr, M=M, N=dim, Cin=8, taps=taps, grid=dict(out_h=H, out_w=W, in_h=H, in_w=W))
    grads["encoder.0.weight"] = dW0.view(dim, 7, 7, 8)[..., :Cin].permute(0, 3, 1, 2).contiguous()
    grads["encoder.0.bias"] = db0
    return grads


class VQVAEForwardFn(torch.autograd.Function):
    """(x_tilde, z_e_x, z_q_x) = VQVAEForwardFn.apply(model, images, names, *params): NCHW outputs as the reference returns them."""

    @staticmethod
    def forward(ctx, vq, x, names, *params):
        with torch.no_grad():
            x_tilde, z_e, zq, tape = (vq_train_forward if vq.down_ratio == 4 else vq8_train_forward)(vq, x)
        N, h, wd, D = tape["N"], tape["h"], tape["wd"], z_e.shape[1]
        ctx.vq, ctx.tape, ctx.names, ctx.shapes = vq, tape, names, [p.shape for p in params]
        return x_tilde, z_e.view(N, h, wd, D).permute(0, 3, 1, 2), zq.view(N, h, wd, D).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g_xt, g_ze, g_zq):
        if ctx.tape is None:
            raise RuntimeError("VQ-VAE training graph: backward through the same forward a second time")
        vq = ctx.vq
        D = vq.dim if vq.down_ratio == 4 else 4 * vq.dim

        def rows(g):                                               # NCHW gradient -> channels-last rows (layout plumbing)
            return None if g is None else g.float().permute(0, 2, 3, 1).reshape(-1, D).contiguous()
        with torch.no_grad(), torch.cuda.device(ctx.tape["x"].device):
            grads = (vq8_train_backward if ctx.tape.get("f8") else vq_train_backward)(vq, ctx.tape, g_xt, rows(g_ze), rows(g_zq))
        ctx.tape = None
        dev = next(vq.parameters()).device
        out = [grads[n].reshape(s) if n in grads else torch.zeros(s, device=dev, dtype=F32) for n, s in zip(ctx.names, ctx.shapes)]
        return (None, None, None, *out)
