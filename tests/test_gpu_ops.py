"""GPU: every libmage_hip.so entry point against a plain PyTorch-CPU fp32 restatement of the same op
(through the C ABI, on seeded inputs).  fp32 mode: 1e-4 class tolerances (exact-fp32 MFMA);
bf16 mode: bf16-rounding class tolerances."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mage_amd import config
from tests.helpers import golden, t

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def ops():
    from mage_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def to_dev(x, dt):
    return x.to(DEV).to(dt).contiguous()


TOL = {torch.float32: dict(atol=2e-5, rtol=2e-5), torch.bfloat16: dict(atol=6e-2, rtol=3e-2)}
# storing a value of magnitude ~1 in a 16-bit type: bf16 keeps 8 significand bits, f16 (MAGE_F16, the single-pass half mode) 11
HTOL = {torch.bfloat16: dict(atol=6e-2, rtol=3e-2), torch.float16: dict(atol=8e-3, rtol=4e-3)}


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (300, 136, 200), (33, 64, 2048), (1000, 1536, 512), (7, 8, 8)])
def test_gemm_plain(dt, M, N, K):
    o = ops()
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    if dt != torch.float32:
        a, w = a.to(dt).float(), w.to(dt).float()
    want = a.double() @ w.double().t() + b.double()
    y = torch.empty(M, N, device=DEV, dtype=torch.float32)
    o.gemm(to_dev(a, dt), to_dev(w, dt), y, M=M, N=N, K=K, lda=K, ldy=N, bias=b.to(DEV))
    torch.testing.assert_close(y.cpu().double(), want, atol=2e-5 if dt == torch.float32 else 2e-3, rtol=1e-5)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(1000, 520, 320), (140000, 520, 320), (140000, 520, 328), (131072 + 8, 512, 64), (4100, 264, 72)])
def test_gemm_lean_epilogue_kinds(dt, M, N, K):
    """The two epilogue kinds without loads (bias [+ act]; x + Linear(.) with the fp32 residual folded into the
    accumulators), bf16 and fp32 output, ragged tiles; (140000, 520, 320) gives every workgroup >= 6 tiles of 256x256 so
    the staggered start of the persistent kernel runs too.  bf16 with K % 64 == 0 and >= 512 tiles runs the 8-phase ping-pong
    kernel (K = 320: 5 slabs; K = 64: a single slab per tile, the DMA cursors two tiles ahead); K = 328 the lockstep one."""
    o = ops()
    a, w, b = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=K ** -0.5), rnd(N, seed=13)
    if dt != torch.float32:
        a, w = a.to(dt).float(), w.to(dt).float()
    lin = a @ w.t() + b
    tol = dict(atol=3e-5, rtol=2e-5) if dt == torch.float32 else dict(atol=3e-3, rtol=1e-4)
    ad, wd, bd = to_dev(a, dt), to_dev(w, dt), b.to(DEV)
    # bias only -> fp32 and bf16
    y = torch.empty(M, N, device=DEV)
    o.gemm(ad, wd, y, M=M, N=N, K=K, lda=K, ldy=N, bias=bd)
    torch.testing.assert_close(y.cpu(), lin, **tol)
    ht = torch.float16 if dt == torch.float16 else torch.bfloat16            # the 16-bit output type: the operands' (fp32 operands write bf16)
    htol = dict(atol=3e-2, rtol=1e-2) if ht == torch.bfloat16 else dict(atol=4e-3, rtol=2e-3)
    yb = torch.empty(M, N, device=DEV, dtype=ht)
    o.gemm(ad, wd, yb, M=M, N=N, K=K, lda=K, ldy=N, bias=bd)
    torch.testing.assert_close(yb.float().cpu(), lin, **htol)
    # QuickGELU -> bf16 (the decoder's c_fc)
    o.gemm(ad, wd, yb, M=M, N=N, K=K, lda=K, ldy=N, bias=bd, act=o.ACT_QUICKGELU)
    torch.testing.assert_close(yb.float().cpu(), lin * torch.sigmoid(1.702 * lin), **htol)
    # x + Linear(.), fp32 residual updated in place (the decoder's out_proj / c_proj)
    res = rnd(M, N, seed=14)
    y = res.to(DEV).clone()
    o.gemm(ad, wd, y, M=M, N=N, K=K, lda=K, ldy=N, bias=bd, residual=y, ldr=N)
    torch.testing.assert_close(y.cpu(), lin + res, **tol)


def test_gemm_8phase_race_screen():
    """The ping-pong kernel orders its LDS traffic with counted vmcnt + barriers only; a misplaced wait shows up as rare
    wrong tiles that come and go with memory load.  Same launch 12 times on three shapes (1, 5 and 32 slabs per tile):
    every run bitwise identical to the first, and the first within bf16 tolerance of the fp32 product."""
    o = ops()
    for M, N, K in ((131072, 512, 64), (140032, 768, 320), (65536, 512, 2048)):
        a, w, b = rnd(M, K, seed=21).bfloat16(), rnd(N, K, seed=22, scale=K ** -0.5).bfloat16(), rnd(N, seed=23)
        ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
        ref = None
        for it in range(12):
            y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            o.gemm(ad, wd, y, M=M, N=N, K=K, lda=K, ldy=N, bias=bd, act=o.ACT_QUICKGELU)
            if it % 3 == 0:                                      # vary what else the memory system is doing
                torch.empty(64 << 20, device=DEV).normal_()
            if ref is None:
                ref = y
                lin = a.float() @ w.float().t() + b
                torch.testing.assert_close(y.float().cpu(), lin * torch.sigmoid(1.702 * lin), atol=3e-2, rtol=1e-2)
            else:
                assert torch.equal(y, ref), f"run {it} of ({M},{N},{K}) differs from run 0"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_epilogue_full(dt):
    """bias -> BN scale/shift -> QuickGELU -> row table -> residual (in place, fp32) -> relu, bf16/fp32 output."""
    o = ops()
    M, N, K, P = 512, 256, 128, 64
    a, w = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5)
    if dt == torch.bfloat16:
        a, w = a.bfloat16().float(), w.bfloat16().float()
    b, sc, sh, tab, res = rnd(N, seed=6), rnd(N, seed=7).abs() + 0.5, rnd(N, seed=8), rnd(P, N, seed=9), rnd(M, N, seed=10)
    v = (a @ w.t() + b) * sc + sh
    v = v * torch.sigmoid(1.702 * v)
    v = v + tab[torch.arange(M) % P] + res
    want = torch.relu(v)
    y = res.to(DEV).clone()
    o.gemm(to_dev(a, dt), to_dev(w, dt), y, M=M, N=N, K=K, lda=K, ldy=N, bias=b.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV),
           act=o.ACT_QUICKGELU, rowadd=tab.to(DEV), rowadd_div=1, rowadd_mod=P, residual=y, ldr=N, post_relu=True)
    torch.testing.assert_close(y.cpu(), want, atol=2e-5 if dt == torch.float32 else 5e-3, rtol=1e-4)
    yb = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    o.gemm(to_dev(a, dt), to_dev(w, dt), yb, M=M, N=N, K=K, lda=K, ldy=N, bias=b.to(DEV), act=o.ACT_GELU_ERF)
    torch.testing.assert_close(yb.float().cpu(), F.gelu(a @ w.t() + b), atol=2e-2, rtol=2e-2)


def test_gemm_row_regrouping():
    """x[:, 1:] views and slot writes without copies (decoder in_linear / head geometry)."""
    o = ops()
    B, L, hw, Cc, N = 3, 5, 16, 64, 32
    x = rnd(B * L * hw, Cc, seed=11)
    w = rnd(N, Cc, seed=12, scale=0.1)
    y = torch.empty(B * (L - 1) * hw, N, device=DEV)
    o.gemm(x.to(DEV), w.to(DEV), y, M=B * (L - 1) * hw, N=N, K=Cc, lda=Cc, ldy=N, out_w=(L - 1) * hw, a_img_stride=L * hw, a_off=hw)
    want = (x.view(B, L, hw, Cc)[:, 1:] @ w.t()).reshape(-1, N)
    torch.testing.assert_close(y.cpu(), want, atol=2e-5, rtol=1e-5)
    tpos = rnd(L, N, seed=13)
    z = torch.zeros(B * L * hw, N, device=DEV)
    imgs = rnd(B * (L - 1) * hw, Cc, seed=14)
    o.gemm(imgs.to(DEV), w.to(DEV), z, M=B * (L - 1) * hw, N=N, K=Cc, lda=Cc, ldy=N, out_w=(L - 1) * hw, y_img_stride=L * hw,
           y_off=hw, rowadd=tpos.to(DEV), rowadd_div=hw, rowadd_mod=L)
    wz = torch.zeros(B, L, hw, N)
    wz[:, 1:] = (imgs @ w.t()).view(B, L - 1, hw, N) + tpos[1:, None, :]
    torch.testing.assert_close(z.cpu(), wz.view(-1, N), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,stride,pad,cin,cout,H", [(3, 1, 1, 64, 64, 16), (4, 2, 1, 32, 48, 32), (1, 1, 0, 40, 24, 8), (3, 1, 1, 8, 8, 16)])
def test_conv_implicit_gemm(dt, k, stride, pad, cin, cout, H):
    o = ops()
    from mage_amd.modules.vqvae_model import VectorQuantizedVAE as V
    n = 3
    x, w, b = rnd(n, cin, H, H, seed=20), rnd(cout, cin, k, k, seed=21, scale=(cin * k * k) ** -0.5), rnd(cout, seed=22)
    if dt == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    want = F.conv2d(x, w, b, stride=stride, padding=pad)
    OH = want.shape[-1]
    xr = x.permute(0, 2, 3, 1).reshape(-1, cin)
    wp = w.permute(0, 2, 3, 1).reshape(cout, -1)
    y = torch.empty(n * OH * OH, cout, device=DEV)
    V._conv(to_dev(xr, dt), to_dev(wp, dt), y, n_img=n, H=H, W=H, cin=cin, cout=cout, k=k, stride=stride, pad=pad, OH=OH, OW=OH,
            bias=b.to(DEV))
    got = y.cpu().view(n, OH, OH, cout).permute(0, 3, 1, 2)
    torch.testing.assert_close(got, want, atol=3e-5 if dt == torch.float32 else 3e-3, rtol=1e-4)


def test_layernorm_both_eps():
    o = ops()
    for Cc, eps in ((512, 1e-5), (64, 1e-8), (1024, 1e-5)):
        x, g, b = rnd(37, Cc, seed=30, scale=3.0) + 0.5, rnd(Cc, seed=31), rnd(Cc, seed=32)
        want = F.layer_norm(x, (Cc,), g, b, eps)
        y = torch.empty(37, Cc, device=DEV)
        o.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), y, eps)
        torch.testing.assert_close(y.cpu(), want, atol=2e-5, rtol=2e-5)
        yb = torch.empty(37, Cc, device=DEV, dtype=torch.bfloat16)
        o.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), yb, eps)
        torch.testing.assert_close(yb.float().cpu(), want, atol=3e-2, rtol=2e-2)


def _ref_attn(q, k, v, H, mask=None):
    R, Tq, E = q.shape
    hd = E // H
    qh, kh, vh = (z.view(R, -1, H, hd).transpose(1, 2) for z in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * hd ** -0.5
    if mask is not None:
        s = s + mask
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(R, Tq, E)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L", [16, 10, 32, 5])
def test_axial_attention_all_axes(dt, L):
    o = ops()
    B, hh, ww, Cc = 2, 16, 16, 64
    H = Cc // 32
    hw = hh * ww
    M = B * L * hw
    qkv = rnd(M, 3 * Cc, seed=40)
    if dt != torch.float32:
        qkv = qkv.to(dt).float()
    x5 = qkv.view(B, L, hh, ww, 3 * Cc)
    dq = to_dev(qkv, dt)
    for axis, geo in ((1, dict(n_seq=B * hw, inner=hw, nq=L, nk=L, q_outer_stride=L * hw, q_axis_stride=hw, causal=True)),
                      (2, dict(n_seq=B * L * ww, inner=ww, nq=hh, nk=hh, q_outer_stride=hw, q_axis_stride=ww, causal=False)),
                      (3, dict(n_seq=B * L * hh, inner=1, nq=ww, nk=ww, q_outer_stride=ww, q_axis_stride=1, causal=False))):
        out = torch.empty(M, Cc, device=DEV, dtype=dt)
        o.attention(dq, dq[:, Cc:], dq[:, 2 * Cc:], out, ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc, ldo=Cc, n_head=H,
                    kv_outer_stride=geo["q_outer_stride"], kv_axis_stride=geo["q_axis_stride"], **geo)
        xt = x5.movedim(axis, -2)
        rows = xt.reshape(-1, xt.shape[-2], 3 * Cc)
        A = rows.shape[1]
        mask = torch.full((A, A), float("-inf")).triu_(1) if axis == 1 else None
        want = _ref_attn(rows[..., :Cc], rows[..., Cc:2 * Cc], rows[..., 2 * Cc:], H, mask)
        want = want.view(*xt.shape[:-1], Cc).movedim(-2, axis).reshape(M, Cc)
        torch.testing.assert_close(out.float().cpu(), want, atol={torch.float32: 2e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dt],
                                   rtol={torch.float32: 1e-4, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dt])


def test_attention_key_padding_and_cross():
    o = ops()
    B, S, Cc, H = 3, 38, 64, 2
    qkv = rnd(B * S, 3 * Cc, seed=41)
    lens = torch.tensor([38, 5, 20], dtype=torch.int32)
    out = torch.empty(B * S, Cc, device=DEV)
    dq = qkv.to(DEV)
    o.attention(dq, dq[:, Cc:], dq[:, 2 * Cc:], out, ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc, ldo=Cc, n_seq=B, inner=1, nq=S, nk=S,
                n_head=H, q_outer_stride=S, q_axis_stride=1, kv_outer_stride=S, kv_axis_stride=1, kv_len=lens.to(DEV))
    r = qkv.view(B, S, 3 * Cc)
    mask = torch.zeros(B, 1, 1, S)
    for b in range(B):
        mask[b, ..., lens[b]:] = float("-inf")
    want = _ref_attn(r[..., :Cc], r[..., Cc:2 * Cc], r[..., 2 * Cc:], H, mask).reshape(B * S, Cc)
    torch.testing.assert_close(out.cpu(), want, atol=2e-5, rtol=1e-4)
    # cross attention, sequence-first addressing (row = t*B + b), 256 queries x 11 keys, 512 channels
    Cc, H, nq, nk = 512, 16, 256, 11
    q, kv = rnd(nq * B, Cc, seed=42), rnd(nk * B, 2 * Cc, seed=43)
    out = torch.empty(nq * B, Cc, device=DEV)
    dkv = kv.to(DEV)
    o.attention(q.to(DEV), dkv, dkv[:, Cc:], out, ldq=Cc, ldk=2 * Cc, ldv=2 * Cc, ldo=Cc, n_seq=B, inner=B, nq=nq, nk=nk, n_head=H,
                q_outer_stride=0, q_axis_stride=B, kv_outer_stride=0, kv_axis_stride=B)
    qb, kvb = q.view(nq, B, Cc).transpose(0, 1), kv.view(nk, B, 2 * Cc).transpose(0, 1)
    want = _ref_attn(qb, kvb[..., :Cc], kvb[..., Cc:], H).transpose(0, 1).reshape(nq * B, Cc)
    torch.testing.assert_close(out.cpu(), want, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("ht", [torch.bfloat16, torch.float16])
def test_attention_mfma_short_sequences_masks(ht):
    """The matrix-core attention kernel (bf16 / f16, nq, nk <= 32: one or two key blocks, query blocks of 16) off the square axial case: a query block appended to a longer key
    cache with the causal mask aligned to the last key (nq = 1 and 3 of nk = 7 and 16: the incremental AR loop), per-sequence
    key lengths, 16 heads, and the same inputs through the vector-ALU kernel (library option attn_no_mfma)."""
    o = ops()
    for nq, nk, causal, use_len in ((1, 7, True, False), (3, 16, True, False), (5, 12, False, True), (16, 16, True, True),
                                    (1, 29, True, False), (32, 32, True, True), (20, 27, False, True), (17, 32, True, False)):
        B, Cc, H = 6, 512, 16
        q = rnd(B * nq, Cc, seed=50 + nq).to(ht)
        kv = rnd(B * nk, 2 * Cc, seed=60 + nk).to(ht)
        lens = torch.tensor([nk, 1, max(1, nk - 3), nk, 2, nk], dtype=torch.int32)
        args = dict(ldq=Cc, ldk=2 * Cc, ldv=2 * Cc, ldo=Cc, n_seq=B, inner=1, nq=nq, nk=nk, n_head=H, q_outer_stride=nq,
                    q_axis_stride=1, kv_outer_stride=nk, kv_axis_stride=1, causal=causal, kv_len=lens.to(DEV) if use_len else None)
        dq, dkv = q.to(DEV), kv.to(DEV)
        out = torch.empty(B * nq, Cc, device=DEV, dtype=ht)
        o.attention(dq, dkv, dkv[:, Cc:], out, **args)
        mask = torch.zeros(B, 1, nq, nk)
        for b in range(B):
            kl = int(lens[b]) if use_len else nk
            for i in range(nq):
                jmax = min(kl, i + 1 + (nk - nq)) if causal else kl
                mask[b, 0, i, jmax:] = float("-inf")
        want = _ref_attn(q.float().view(B, nq, Cc), kv.float().view(B, nk, 2 * Cc)[..., :Cc], kv.float().view(B, nk, 2 * Cc)[..., Cc:], H,
                         mask).reshape(B * nq, Cc)
        torch.testing.assert_close(out.float().cpu(), want, atol=2e-2 if ht == torch.bfloat16 else 3e-3, rtol=2e-2 if ht == torch.bfloat16 else 3e-3)
        with config.lib_option("attn_no_mfma", 1):
            out2 = torch.empty_like(out)
            o.attention(dq, dkv, dkv[:, Cc:], out2, **args)
        torch.testing.assert_close(out2.float().cpu(), out.float().cpu(), atol=1e-2 if ht == torch.bfloat16 else 2e-3, rtol=1e-2)


def test_vq_nearest_golden_ties_and_margin():
    o = ops()
    g = golden("vq_unit")
    for zk, ck, ik in (("z", "cb", "idx"), ("z2", "cb2", "idx2")):
        z, cb = t(g[zk]).to(DEV), t(g[ck]).to(DEV)
        cbt, c2 = o.vq_prepare(cb)
        idx, margin = o.vq_nearest(z.reshape(-1, cb.shape[1]).contiguous(), cbt, c2, want_margin=True)
        want = t(g[ik]).reshape(-1)
        bad = idx.cpu() != want
        if zk == "z":
            assert not bad.any()                              # exact ties: lowest index wins, like torch.min
            assert margin[0].item() == 0.0 and margin[1].item() == 0.0
        else:                                                  # reference-init regime: distances quantised at ulp(|z|^2)
            m = t(g["margin2"]).reshape(-1)
            assert not (bad & (m > 1e-5)).any()


def test_embedding_argmax_ce_rowaffine():
    o = ops()
    table = rnd(50, 64, seed=50)
    ids = torch.randint(0, 50, (7, 9), generator=torch.Generator().manual_seed(1))
    out = torch.empty(63, 64, device=DEV)
    o.embedding(ids.to(DEV), table.to(DEV), out, relu=True)
    torch.testing.assert_close(out.cpu(), torch.relu(table[ids.reshape(-1)]))
    logits = rnd(40, 512, seed=51)
    logits[3, 100] = logits[3, 7] = 9.0                       # tie: first maximum
    tgt = torch.randint(0, 512, (40,), generator=torch.Generator().manual_seed(2))
    am = torch.empty(40, device=DEV, dtype=torch.int64)
    mg = torch.empty(40, device=DEV)
    o.argmax(logits.to(DEV), am, rows=40, K=512, margin=mg)
    assert torch.equal(am.cpu(), logits.max(-1)[1]) and am[3].item() == 7 and mg[3].item() == 0.0
    # regrouped: pick frame 2 of [B=4, T=5, hw=2, K] and write into slot 3 of a [B, 5, hw] token buffer
    lg = rnd(4 * 5 * 2, 512, seed=52)
    buf = torch.full((4, 5, 2), -1, device=DEV, dtype=torch.int64)
    o.argmax(lg.to(DEV), buf, rows=8, K=512, group=2, in_group_stride=10, in_off=4, out_group_stride=10, out_off=6)
    wantb = torch.full((4, 5, 2), -1, dtype=torch.int64)
    wantb[:, 3] = lg.view(4, 5, 2, 512)[:, 2].max(-1)[1]
    assert torch.equal(buf.cpu(), wantb)
    loss = o.cross_entropy(logits.to(DEV), tgt.to(DEV))
    assert abs(loss.item() - F.cross_entropy(logits, tgt).item()) < 1e-5
    x, rs, tab = rnd(24, 64, seed=53), rnd(24, seed=54), rnd(6, 64, seed=55)
    dx = x.to(DEV).clone()
    o.row_affine(dx, rs.to(DEV), tab.to(DEV), div=2, mod=6)
    torch.testing.assert_close(dx.cpu(), x * rs[:, None] + tab[(torch.arange(24) // 2) % 6])


def test_direct_convs_pool_upsample_adain():
    o = ops()
    # f4 stem: Conv2d(1, 64, 4, 2, 1) + BN + ReLU, NCHW image -> channels-last
    x, w, b = rnd(2, 1, 64, 64, seed=60), rnd(64, 1, 4, 4, seed=61, scale=0.25), rnd(64, seed=62)
    sc, sh = rnd(64, seed=63).abs() + 0.5, rnd(64, seed=64)
    want = torch.relu((F.conv2d(x, w, b, stride=2, padding=1)) * sc[None, :, None, None] + sh[None, :, None, None])
    y = torch.empty(2 * 32 * 32, 64, device=DEV)
    o.conv_in(x.to(DEV), w.permute(1, 2, 3, 0).contiguous().to(DEV), b.to(DEV), sc.to(DEV), sh.to(DEV), y, cin=1, H=64, W=64,
              cout=64, kh=4, kw=4, stride=2, pad=1, act=o.ACT_RELU)
    torch.testing.assert_close(y.cpu().view(2, 32, 32, 64).permute(0, 3, 1, 2), want, atol=2e-5, rtol=1e-5)
    # many frames (>= 65536 output pixels, <= 16 taps): persistent waves with the weights in registers and scalar tap loads -- the same
    # bits as the per-pixel kernel that smaller calls take; plain fp32 rows (64 and 256 channels) and the split space-to-depth layout
    for cout_, ydt in ((64, "f32"), (256, "f32"), (256, "s2d")):
        xb = rnd(64, 1, 64, 64, seed=160).to(DEV)
        wb = rnd(cout_, 1, 4, 4, seed=161, scale=0.25).permute(1, 2, 3, 0).contiguous().to(DEV)
        bb, scb, shb = rnd(cout_, seed=162).to(DEV), (rnd(cout_, seed=163).abs() + 0.5).to(DEV), rnd(cout_, seed=164).to(DEV)
        kwc = dict(cin=1, H=64, W=64, cout=cout_, kh=4, kw=4, stride=2, pad=1, act=o.ACT_RELU)
        if ydt == "f32":
            big = torch.empty(64 * 32 * 32, cout_, device=DEV)
            o.conv_in(xb, wb, bb, scb, shb, big, **kwc)
            parts = torch.empty_like(big)
            for s0 in (0, 32):
                o.conv_in(xb[s0:s0 + 32].contiguous(), wb, bb, scb, shb, parts[s0 * 1024:(s0 + 32) * 1024], **kwc)
        else:
            big = o.split_empty(64 * 17 * 17 + 1, 4 * cout_, o.F16X3, DEV, zero=True)
            o.conv_in(xb, wb, bb, scb, shb, big, split_kind=o.F16X3, s2d=True, **kwc)
            parts = o.split_empty(64 * 17 * 17 + 1, 4 * cout_, o.F16X3, DEV, zero=True)
            for s0 in (0, 32):
                o.conv_in(xb[s0:s0 + 32].contiguous(), wb, bb, scb, shb, parts[s0 * 289:], split_kind=o.F16X3, s2d=True, **kwc)
        assert torch.equal(big.float(), parts.float())              # as values: a tap outside the image adds 0 * w here, nothing there (-0 / +0)
    # f8 stem: Conv2d(3, 32, 7, padding=3)
    x, w, b = rnd(1, 3, 32, 32, seed=65), rnd(32, 3, 7, 7, seed=66, scale=0.1), rnd(32, seed=67)
    y = torch.empty(32 * 32, 32, device=DEV)
    o.conv_in(x.to(DEV), w.permute(1, 2, 3, 0).contiguous().to(DEV), b.to(DEV), None, None, y, cin=3, H=32, W=32, cout=32, kh=7,
              kw=7, stride=1, pad=3)
    torch.testing.assert_close(y.cpu().view(1, 32, 32, 32).permute(0, 3, 1, 2), F.conv2d(x, w, b, padding=3), atol=3e-5, rtol=1e-5)
    # ConvTranspose2d(64, 1, 4, 2, 1) + tanh and Conv2d(64, 3, 1) + tanh, channels-last -> NCHW
    xi, wt, bt = rnd(2, 64, 8, 8, seed=68), rnd(64, 1, 4, 4, seed=69, scale=0.1), rnd(1, seed=70)
    out = torch.empty(2, 1, 16, 16, device=DEV)
    o.conv_out(xi.permute(0, 2, 3, 1).contiguous().to(DEV), wt.permute(2, 3, 1, 0).contiguous().to(DEV), bt.to(DEV), out, N=2, IH=8,
               IW=8, cin=64, cout=1, transposed=True)
    torch.testing.assert_close(out.cpu(), torch.tanh(F.conv_transpose2d(xi, wt, bt, stride=2, padding=1)), atol=2e-5, rtol=1e-5)
    # the same transposed head as GEMM (per-input-pixel tap products) + fold, 1 and 3 output channels
    for cout, seed in ((1, 69), (3, 169)):
        wt = rnd(64, cout, 4, 4, seed=seed, scale=0.1)
        bt = rnd(cout, seed=seed + 1)
        a = xi.permute(0, 2, 3, 1).contiguous().view(2 * 64, 64).to(DEV)
        w16 = wt.permute(2, 3, 1, 0).contiguous().view(16 * cout, 64).to(DEV)            # row (ky*4+kx)*cout + co
        taps = torch.empty(2 * 64, 16 * cout, device=DEV)
        o.gemm(a, w16, taps, M=128, N=16 * cout, K=64, lda=64, ldy=16 * cout)
        out = torch.empty(2, cout, 16, 16, device=DEV)
        o.convt_fold_tanh(taps, bt.to(DEV), out, N=2, IH=8, IW=8, cout=cout)
        torch.testing.assert_close(out.cpu(), torch.tanh(F.conv_transpose2d(xi, wt, bt, stride=2, padding=1)), atol=2e-5, rtol=1e-5)
    # many images, one output channel: the fold runs one image per workgroup through LDS -- the same bits as the per-pixel kernel
    # (which fewer than 64 images take), borders included
    for IH, IW in ((32, 32), (6, 10)):
        tp = rnd(70 * IH * IW, 16, seed=171).to(DEV)
        bt = rnd(1, seed=172).to(DEV)
        big = torch.empty(70, 1, 2 * IH, 2 * IW, device=DEV)
        o.convt_fold_tanh(tp, bt, big, N=70, IH=IH, IW=IW, cout=1)
        parts = torch.empty_like(big)
        for s0 in (0, 35):
            o.convt_fold_tanh(tp[s0 * IH * IW:(s0 + 35) * IH * IW], bt, parts[s0:s0 + 35], N=35, IH=IH, IW=IW, cout=1)
        assert torch.equal(big, parts)
    w1, b1 = rnd(3, 64, 1, 1, seed=71, scale=0.2), rnd(3, seed=72)
    out = torch.empty(2, 3, 8, 8, device=DEV)
    o.conv_out(xi.permute(0, 2, 3, 1).contiguous().to(DEV), w1.reshape(3, 64).contiguous().to(DEV), b1.to(DEV), out, N=2, IH=8, IW=8,
               cin=64, cout=3, transposed=False)
    torch.testing.assert_close(out.cpu(), torch.tanh(F.conv2d(xi, w1, b1)), atol=2e-5, rtol=1e-5)
    # pool / upsample / relu / cast
    xc = rnd(2, 8, 8, 16, seed=73)
    p = torch.empty(2 * 4 * 4, 16, device=DEV)
    o.maxpool2(xc.to(DEV), p, N=2, H=8, W=8, Cc=16)
    torch.testing.assert_close(p.cpu().view(2, 4, 4, 16), F.max_pool2d(xc.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))
    u = torch.empty(2 * 16 * 16, 16, device=DEV)
    o.upsample2(xc.to(DEV), u, N=2, H=8, W=8, Cc=16)
    torch.testing.assert_close(u.cpu().view(2, 16, 16, 16), F.interpolate(xc.permute(0, 3, 1, 2), scale_factor=2).permute(0, 2, 3, 1))
    r = o.relu(xc.to(DEV), torch.empty_like(xc, device=DEV))
    torch.testing.assert_close(r.cpu(), torch.relu(xc))
    cb = o.cast(xc.to(DEV), torch.empty(xc.shape, device=DEV, dtype=torch.bfloat16))
    assert torch.equal(cb.cpu(), xc.bfloat16())
    # ADAIN + speed add
    B, P, Cc = 2, 256, 64
    xm, gm, bt2 = rnd(B, P, Cc, seed=74, scale=2.0) + 1, rnd(B, P, Cc, seed=75), rnd(B, P, Cc, seed=76)
    oa = o.adain(xm.to(DEV), gm.to(DEV), bt2.to(DEV), torch.empty(B, P, Cc, device=DEV), B=B, P=P, Cc=Cc)
    want = gm * F.instance_norm(xm.permute(0, 2, 1)).permute(0, 2, 1) + bt2
    torch.testing.assert_close(oa.cpu(), want, atol=2e-5, rtol=1e-4)
    sp, vec = rnd(B, seed=77), rnd(Cc, seed=78)
    xs = xm.to(DEV).clone()
    o.add_scaled_rowvec(xs, sp.to(DEV), vec.to(DEV), B=B, P=P, Cc=Cc)
    torch.testing.assert_close(xs.cpu(), xm + sp[:, None, None] * vec)


def test_errors_are_loud():
    o = ops()
    a = torch.zeros(8, 8, device=DEV)
    with pytest.raises(ValueError):
        o.gemm(a, a, a, M=8, N=4, K=8, lda=8, ldy=4)           # N not a multiple of 8
    with pytest.raises(RuntimeError):
        o.layernorm(torch.zeros(4, 8), torch.zeros(8), torch.zeros(8), torch.zeros(4, 8), 1e-5)   # CPU tensors: no fallback


def test_out_of_range_ids_and_targets_are_reported():
    """nn.Embedding raises IndexError on an id outside the table and F.cross_entropy on a bad target; the kernels record the
    event in the device's deferred-error word and mage_check_device_errors turns it into ValueError (and clears it)."""
    o = ops()
    table = rnd(10, 16, seed=90).to(DEV)
    ids = torch.tensor([0, 3, 9], device=DEV)
    o.embedding(ids, table, torch.empty(3, 16, device=DEV))
    o.check_device_errors(DEV)                                   # clean
    bad = torch.tensor([0, 10, -1], device=DEV)
    out = o.embedding(bad, table, torch.empty(3, 16, device=DEV))
    with pytest.raises(ValueError, match="index out of range"):
        o.check_device_errors(DEV)
    o.check_device_errors(DEV)                                   # cleared by the report
    assert torch.isfinite(out).all()                             # and the launch stayed memory-safe
    logits = rnd(8, 32, seed=91).to(DEV)
    o.cross_entropy(logits, torch.full((8,), 31, device=DEV, dtype=torch.int64))
    o.check_device_errors(DEV)
    o.cross_entropy(logits, torch.full((8,), 32, device=DEV, dtype=torch.int64))
    with pytest.raises(ValueError, match="target out of range"):
        o.check_device_errors(DEV)


@pytest.mark.parametrize("n_img,R,half,padded,relu", [(64, 32, False, False, True), (64, 32, True, False, True), (16, 64, True, True, True),
                                                      (16, 128, False, True, False), (300, 16, False, False, True)])
def test_tile_convolution_64_channels_equals_the_implicit_gemm(n_img, R, half, padded, relu):
    """conv3x3 64 -> 64 channels over whole 16 x 16 pixel tiles (conv_tile.hip: the 18 x 18 input window and the weights resident in LDS) against
    the implicit-GEMM kernel on the same descriptor (library option conv_no_tile): the same accumulation chains, bit for bit; and against
    torch's conv2d on the bf16-rounded operands.  half: the input at half resolution (nn.Upsample folded into the fetch); padded: the rows
    written into the interior of a zero-padded frame buffer (the next padded-taps convolution's input)."""
    import torch.nn.functional as F
    from mage_amd import config
    o = ops()
    Ri = R // 2 if half else R
    x = rnd(n_img, Ri, Ri, 64, seed=130).bfloat16()
    w = rnd(64, 3, 3, 64, seed=131, scale=(9 * 64) ** -0.5).bfloat16()                  # [co, ky, kx, ci]
    bias = rnd(64, seed=132, scale=0.1)
    xin = x.float().permute(0, 3, 1, 2)
    if half:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    if relu:
        ref = torch.relu(ref)
    P = R + 2
    geo = dict(y_img_stride=P * P, y_mul_y=P, y_off=P + 1) if padded else {}
    kw = dict(M=n_img * R * R, N=64, K=576, lda=64, ldy=64, out_h=R, out_w=R, in_h=R, in_w=R, taps_h=3, taps_w=3, cin=64, stride=1, dy0=-1, dx0=-1,
              bias=bias.to(DEV), act=o.ACT_RELU if relu else o.ACT_NONE, **geo)
    if half:
        kw.update(a_half=True, a_img_stride=Ri * Ri)
    xd, wd = x.view(-1, 64).to(DEV), w.reshape(64, 576).to(DEV)
    rows = n_img * P * P + 1 if padded else n_img * R * R
    y_tile = torch.zeros(rows, 64, device=DEV, dtype=torch.bfloat16)
    y_gemm = torch.zeros_like(y_tile)
    o.gemm(xd, wd, y_tile, **kw)
    with config.lib_option("conv_no_tile", 1):
        o.gemm(xd, wd, y_gemm, **kw)
    assert torch.equal(y_tile, y_gemm)
    got = y_tile[:n_img * P * P].view(n_img, P, P, 64)[:, 1:-1, 1:-1] if padded else y_tile.view(n_img, R, R, 64)
    torch.testing.assert_close(got.float().cpu(), ref, atol=3e-2, rtol=2e-2)
    if padded:                                                  # the padding ring stays zero
        full = y_tile[:n_img * P * P].view(n_img, P, P, 64)
        assert full[:, 0].abs().max().item() == 0 and full[:, :, 0].abs().max().item() == 0 and full[:, -1].abs().max().item() == 0


@pytest.mark.parametrize("n_img,R,hid,cout,half", [(32, 32, 64, 256, True), (16, 64, 64, 256, False), (64, 16, 128, 512, False)])
def test_padded_taps_conv_adds_a_residual_tensor_in_its_epilogue(n_img, R, hid, cout, half):
    """A bottleneck block's closing convolution (3x3, hid -> cout) + identity path in the padded-taps form of the 8-phase kernel: the bf16 identity
    rows -- at half resolution behind an Upsample (res_half) -- are fetched in the epilogue.  Against the implicit-GEMM kernel's general
    epilogue on the unpadded input: with 64 hidden channels (a K slab is one tap) the same bits; with 128 the same sums in another order."""
    o = ops()
    P = R + 2
    x = rnd(n_img, R, R, hid, seed=140).bfloat16()
    w = rnd(cout, 3, 3, hid, seed=141, scale=(9 * hid) ** -0.5).bfloat16()
    bias = rnd(cout, seed=142, scale=0.1).to(DEV)
    Rr = R // 2 if half else R
    res = rnd(n_img * Rr * Rr, cout, seed=143).bfloat16().to(DEV)
    pad = torch.zeros(n_img, P, P, hid, dtype=torch.bfloat16)
    pad[:, 1:-1, 1:-1] = x
    pad = torch.cat([pad.view(-1, hid), torch.zeros(1, hid, dtype=torch.bfloat16)]).to(DEV)
    wd = w.reshape(cout, 9 * hid).to(DEV)
    M = n_img * R * R
    y_taps = torch.empty(M, cout, device=DEV, dtype=torch.bfloat16)
    y_gen = torch.empty_like(y_taps)
    o.gemm(pad, wd, y_taps, M=M, N=cout, K=9 * hid, lda=hid, ldy=cout, out_h=R, out_w=R, in_h=P, in_w=P, a_img_stride=P * P, taps_h=3, taps_w=3,
           cin=hid, stride=1, dy0=0, dx0=0, bias=bias, residual=res, ldr=cout, res_half=half)
    o.gemm(x.view(-1, hid).to(DEV), wd, y_gen, M=M, N=cout, K=9 * hid, lda=hid, ldy=cout, out_h=R, out_w=R, in_h=R, in_w=R, taps_h=3, taps_w=3,
           cin=hid, stride=1, dy0=-1, dx0=-1, bias=bias, residual=res, ldr=cout, res_half=half)
    if hid == 64:
        assert torch.equal(y_taps, y_gen)
    else:
        assert (y_taps.float() - y_gen.float()).abs().max().item() <= 2 ** -6 * max(1.0, y_gen.float().abs().max().item())
    import torch.nn.functional as F
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1) + bias.cpu()
    r4 = res.float().cpu().view(n_img, Rr, Rr, cout)
    if half:
        r4 = r4.repeat_interleave(2, 1).repeat_interleave(2, 2)
    torch.testing.assert_close(y_taps.float().cpu().view(n_img, R, R, cout), ref + r4, atol=4e-2, rtol=2e-2)


def test_gemm_over_relu_of_the_operand_rows():
    """mage_gemm_desc::a_relu (the 256 x 64 tile, bf16): A W^T over relu(A) == the product over a stored relu(A), bit for bit; refused
    where the shape does not run on that tile."""
    o = ops()
    M, K = 256 * 256, 256
    for N in (64, 128):
        a = rnd(M, K, seed=120).bfloat16().to(DEV)
        w = rnd(N, K, seed=121, scale=K ** -0.5).bfloat16().to(DEV)
        bias = rnd(N, seed=122, scale=0.1).to(DEV)
        y0 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        y1 = torch.empty_like(y0)
        o.gemm(torch.relu(a), w, y0, M=M, N=N, K=K, lda=K, ldy=N, bias=bias, act=o.ACT_RELU)
        o.gemm(a, w, y1, M=M, N=N, K=K, lda=K, ldy=N, bias=bias, act=o.ACT_RELU, a_relu=True)
        assert torch.equal(y0, y1)
        ref = torch.relu(torch.relu(a.float()) @ w.float().t() + bias)
        torch.testing.assert_close(y1.float(), ref, atol=3e-2, rtol=2e-2)
    with pytest.raises(Exception, match="a_relu"):             # 256 columns: not the narrow tile
        w2 = rnd(256, K, seed=123).bfloat16().to(DEV)
        o.gemm(a, w2, torch.empty(M, 256, device=DEV, dtype=torch.bfloat16), M=M, N=256, K=K, lda=K, ldy=256, a_relu=True)
    with pytest.raises(Exception, match="a_relu"):             # too few rows for a tile per CU
        o.gemm(a, w, y1, M=4096, N=N, K=K, lda=K, ldy=N, bias=bias, a_relu=True)


@pytest.mark.parametrize("ht", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n_img,Cc,mode", [(8, 256, "table"), (3, 512, "table"), (6, 256, "relu_bias"), (5, 256, "plain")])
def test_padded_taps_conv_on_the_8phase_kernel(n_img, Cc, mode, ht):
    """conv3x3 over a zero-padded frame buffer (every tap a valid row): the padded-taps form of mage_gemm runs on the 8-phase
    ping-pong kernel (one scalar offset per K slab); y = table[row % 256] + conv (the frame convolution + H/W positions), or
    act(conv + bias).  Against torch's conv2d on the same bf16-rounded operands, and against the generic gather kernel's result."""
    import torch.nn.functional as F
    o = ops()
    if ht == torch.float16 and mode != "table":
        pytest.skip("MAGE_F16 takes the row-table form of the padded-taps kernel (the decoder stream's fills); the VQ-VAE forms are bf16")
    R, P = 16, 18
    x = rnd(n_img, R, R, Cc, seed=70).to(ht)
    w = rnd(Cc, 3, 3, Cc, seed=71, scale=(9 * Cc) ** -0.5).to(ht)                      # [co, ky, kx, ci]
    table = rnd(R * R, Cc, seed=72)
    bias = rnd(Cc, seed=73, scale=0.1)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(n_img * R * R, Cc)
    kw = {}
    if mode == "table":
        ref = ref + table.repeat(n_img, 1)
        kw = dict(rowadd=table.to(DEV), rowadd_div=1, rowadd_mod=R * R)
    elif mode == "relu_bias":
        ref = torch.relu(ref + bias)
        kw = dict(bias=bias.to(DEV), act=o.ACT_RELU)
    pad = torch.zeros(n_img, P, P, Cc, dtype=ht)
    pad[:, 1:-1, 1:-1] = x
    wd = w.reshape(Cc, 9 * Cc).to(DEV)
    y = torch.empty(n_img * R * R, Cc, device=DEV, dtype=ht)
    o.gemm(pad.view(-1, Cc).to(DEV), wd, y, M=n_img * R * R, N=Cc, K=9 * Cc, lda=Cc, ldy=Cc, out_h=R, out_w=R, in_h=P, in_w=P,
           a_img_stride=P * P, taps_h=3, taps_w=3, cin=Cc, stride=1, dy0=0, dx0=0, **kw)
    torch.testing.assert_close(y.float().cpu(), ref, atol=3e-2 if ht == torch.bfloat16 else 4e-3, rtol=2e-2 if ht == torch.bfloat16 else 3e-3)
    # the generic gather kernel on the unpadded input: the same sums in another order, rounded to bf16
    y2 = torch.empty_like(y)
    o.gemm(x.view(-1, Cc).to(DEV), wd, y2, M=n_img * R * R, N=Cc, K=9 * Cc, lda=Cc, ldy=Cc, out_h=R, out_w=R, in_h=R, in_w=R, taps_h=3,
           taps_w=3, cin=Cc, stride=1, dy0=-1, dx0=-1, **kw)
    assert (y.float() - y2.float()).abs().max().item() <= 2 ** (-6 if ht == torch.bfloat16 else -9) * max(1.0, ref.abs().max().item())
    # and the embedding kernel's two-level addressing fills exactly the interior of the padded buffer
    ids = torch.randint(0, 50, (n_img * R * R,), generator=torch.Generator().manual_seed(74))
    tab = rnd(50, Cc, seed=75)
    buf = torch.zeros(n_img * P * P, Cc, device=DEV, dtype=ht)
    o.embedding(ids.to(DEV), tab.to(DEV), buf, group=R * R, group_stride=P * P, off=P + 1, inner=R, inner_stride=P)
    want = torch.zeros(n_img, P, P, Cc, dtype=ht)
    want[:, 1:-1, 1:-1] = tab[ids].to(ht).view(n_img, R, R, Cc)
    assert torch.equal(buf.cpu().view(n_img, P, P, Cc), want)


@pytest.mark.parametrize("n_img", [2, 5, 264])
def test_padded_taps_gemm_with_the_narrow_head_on_its_tile(n_img):
    """mage_gemm_desc::head_w: the sub-pixel GEMMs of ConvTranspose2d(256, 256, 4, 2, 1) + BN + ReLU (vqvae_model.py:184-186; 2 x 2 taps over
    the padded frame buffer, rows interleaved into the 2x grid) with the last ConvTranspose2d's 16 taps (:187) taken on each tile's
    bf16-rounded rows before they leave the CU.  Three checks: (1) against fp64 sums over the SAME bf16 rows the unfused kernel stores
    (the fusion changes the summation order of the head only: fp32-sum tolerance); (2) rows the launch does not own keep their bits
    (the other sub-pixel phases' slots); (3) a second launch gives the same bits (fixed-order reduction over the four wave columns).
    n_img = 264: more than one tile per workgroup on 256 CUs -- the staging windows are reused under the next tile's loads."""
    o = ops()
    R, P, Cc = 16, 18, 256
    x = torch.relu(rnd(n_img, R, R, Cc, seed=170)).bfloat16()
    w = rnd(Cc, 2, 2, Cc, seed=171, scale=(4 * Cc) ** -0.5).bfloat16()                     # [co, ky, kx, ci]
    bias = rnd(Cc, seed=172, scale=0.1)
    hw = rnd(16, Cc, seed=173, scale=Cc ** -0.5).bfloat16()
    pad = torch.zeros(n_img * P * P + 1, Cc, dtype=torch.bfloat16)
    pad[:-1].view(n_img, P, P, Cc)[:, 1:-1, 1:-1] = x
    pad_d, wd, hwd, bd = pad.to(DEV), w.reshape(Cc, 4 * Cc).to(DEV), hw.to(DEV), bias.to(DEV)
    hwp = R * R
    M = n_img * hwp
    win = dict(out_h=R, out_w=R, in_h=P, in_w=P, a_img_stride=P * P, cin=Cc, stride=1, dy0=0, dx0=0, taps_h=2, taps_w=2)
    for py, px in ((0, 0), (1, 1)):
        geo = dict(a_off=py * P + px, y_img_stride=4 * hwp, y_mul_y=4 * R, y_mul_x=2, y_off=py * 2 * R + px)
        up = torch.zeros(4 * M, Cc, device=DEV, dtype=torch.bfloat16)
        o.gemm(pad_d, wd, up, M=M, N=Cc, K=4 * Cc, lda=Cc, ldy=Cc, bias=bd, act=o.ACT_RELU, **geo, **win)
        taps = torch.full((4 * M, 16), -7.0, device=DEV)
        o.gemm(pad_d, wd, taps, M=M, N=Cc, K=4 * Cc, lda=Cc, ldy=16, bias=bd, act=o.ACT_RELU, head_w=hwd, **geo, **win)
        upc, tc = up.cpu(), taps.cpu()
        rows = (torch.arange(n_img)[:, None, None] * 4 * hwp + torch.arange(R)[None, :, None] * 4 * R + torch.arange(R)[None, None, :] * 2
                + py * 2 * R + px).reshape(-1)
        want = (upc[rows].double() @ hw.double().t())
        got = tc[rows].double()
        assert upc[rows].abs().max().item() > 0.5                                            # not a test of zeros
        scale = (upc[rows].double().abs() @ hw.double().abs().t()).clamp_min(1e-3)         # the size of the sums that were added up
        assert ((got - want).abs() / scale).max().item() < 2e-6, ((got - want).abs() / scale).max().item()
        other = torch.ones(4 * M, dtype=torch.bool)
        other[rows] = False
        assert torch.all(tc[other] == -7.0)
        taps2 = torch.full((4 * M, 16), -7.0, device=DEV)
        o.gemm(pad_d, wd, taps2, M=M, N=Cc, K=4 * Cc, lda=Cc, ldy=16, bias=bd, act=o.ACT_RELU, head_w=hwd, **geo, **win)
        assert torch.equal(taps2.cpu(), tc)
    # the four phases in ONE launch (head_phases = 4: stacked weights, column tile = phase): the four launches' bits
    w4 = [rnd(Cc, 2, 2, Cc, seed=180 + i, scale=(4 * Cc) ** -0.5).bfloat16().reshape(Cc, 4 * Cc).to(DEV) for i in range(4)]
    four = torch.full((4 * M, 16), -7.0, device=DEV)
    for i, (py, px) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        o.gemm(pad_d, w4[i], four, M=M, N=Cc, K=4 * Cc, lda=Cc, ldy=16, bias=bd, act=o.ACT_RELU, head_w=hwd, a_off=py * P + px, y_img_stride=4 * hwp,
               y_mul_y=4 * R, y_mul_x=2, y_off=py * 2 * R + px, **win)
    one = torch.full((4 * M, 16), -7.0, device=DEV)
    o.gemm(pad_d, torch.cat(w4, 0).contiguous(), one, M=M, N=4 * Cc, K=4 * Cc, lda=Cc, ldy=16, bias=bd.repeat(4).contiguous(), act=o.ACT_RELU, head_w=hwd,
           head_phases=4, a_off=0, y_img_stride=4 * hwp, y_mul_y=4 * R, y_mul_x=2, y_off=0, **win)
    assert torch.equal(one.cpu(), four.cpu()) and not (one == -7.0).any()
    # a geometry the fusion does not cover is refused, not silently run unfused
    with pytest.raises(Exception, match="head_w"):
        o.gemm(pad_d, wd, taps, M=M, N=Cc, K=4 * Cc, lda=Cc, ldy=16, bias=bd, act=o.ACT_NONE, head_w=hwd, **geo, **win)


@pytest.mark.parametrize("ht", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [512, 2048, 32768])
def test_layernorm_folded_around_the_gemms(M, ht):
    """bf16: the x + Linear(.) GEMM that also writes a bf16 copy of x and per-row partial (sum, sum of squares); mage_ln_stats;
    the Linear that consumes (copy, stats) with gamma folded into its weights = Linear(LayerNorm(x)) (mage_model.py:35-53).
    M = 512 runs the few-rows kernel (gemm_small_kernel), M = 2048 the lockstep one, M = 32768 the 8-phase one; the same rows give
    the same bits on all of them (the 256-row slices below go through the few-rows kernel)."""
    o = ops()
    C_, eps = 1024, 1e-5
    x0 = rnd(M, C_, seed=1) + 0.3
    ao = rnd(M, C_, seed=2).to(ht)
    wo, bo = rnd(C_, C_, seed=3, scale=C_ ** -0.5).to(ht), rnd(C_, seed=4, scale=0.1)
    g, bt = 1.0 + 0.2 * rnd(C_, seed=5), 0.1 * rnd(C_, seed=6)
    wf, bf = rnd(4 * C_, C_, seed=7, scale=C_ ** -0.5), rnd(4 * C_, seed=8, scale=0.1)
    xd, aod, wod, bod = x0.to(DEV), ao.to(DEV), wo.to(DEV), bo.to(DEV)
    # producer against the same GEMM without the extra outputs: the fp32 stream must not change by a bit
    x_plain = xd.clone()
    o.gemm(aod, wod, x_plain, M=M, N=C_, K=C_, lda=C_, ldy=C_, bias=bod, residual=x_plain, ldr=C_)
    x_new = xd.clone()
    xb = torch.empty(M, C_, device=DEV, dtype=ht)
    part = torch.empty(C_ // 64, M, 2, device=DEV, dtype=torch.float32)             # slice-major (mage_gemm_desc::ln_part_rows)
    o.gemm(aod, wod, x_new, M=M, N=C_, K=C_, lda=C_, ldy=C_, bias=bod, residual=x_new, ldr=C_, y2=xb, ldy2=C_, ln_part=part)
    assert torch.equal(x_new, x_plain)
    assert torch.equal(xb, x_new.to(ht))
    xs = x_new.double().view(M, C_ // 64, 64)
    torch.testing.assert_close(part[..., 0].t().double(), xs.sum(-1), atol=1e-4, rtol=1e-5)
    torch.testing.assert_close(part[..., 1].t().double(), (xs * xs).sum(-1), atol=1e-3, rtol=1e-5)
    stats = torch.empty(M, 2, device=DEV, dtype=torch.float32)
    o.ln_stats(part, C_, eps, stats)
    mean = x_new.double().mean(-1)
    var = x_new.double().var(-1, unbiased=False)
    torch.testing.assert_close(stats[:, 0].double(), mean, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(stats[:, 1].double(), (var + eps).rsqrt(), atol=1e-4, rtol=1e-4)
    # consumer
    wq = (wf * g[None, :]).to(ht)
    s = wq.float().sum(1)
    c = (wf.double() @ bt.double() + bf.double()).float()
    want = F.layer_norm(x_new.cpu().double(), (C_,), g.double(), bt.double(), eps) @ wf.double().t() + bf.double()
    for act in (o.ACT_NONE, o.ACT_QUICKGELU):
        y = torch.empty(M, 4 * C_, device=DEV, dtype=ht)
        o.gemm(xb, wq.to(DEV), y, M=M, N=4 * C_, K=C_, lda=C_, ldy=4 * C_, bias=c.to(DEV), act=act, ln_stats=stats,
               ln_colsum=s.to(DEV))
        w_ = want if act == o.ACT_NONE else want * torch.sigmoid(1.702 * want)
        torch.testing.assert_close(y.cpu().double(), w_, **HTOL[ht])
        # the same rows through the other kernel (fewer rows: lockstep 128-row tiles): bit-identical
        y_s = torch.empty(256, 4 * C_, device=DEV, dtype=ht)
        o.gemm(xb[:256], wq.to(DEV), y_s, M=256, N=4 * C_, K=C_, lda=C_, ldy=4 * C_, bias=c.to(DEV), act=act, ln_stats=stats[:256],
               ln_colsum=s.to(DEV))
        assert torch.equal(y_s, y[:256])
        # ... and with (mean, rstd) taken from the partial sums inside the few-rows kernel (no mage_ln_stats launch): the same bits
        assert o.gemm_is_small(xb, 256, 4 * C_, C_)
        y_p = torch.empty(256, 4 * C_, device=DEV, dtype=ht)
        o.gemm(xb[:256], wq.to(DEV), y_p, M=256, N=4 * C_, K=C_, lda=C_, ldy=4 * C_, bias=c.to(DEV), act=act, ln_part=part,
               ln_eps=eps, ln_colsum=s.to(DEV))             # (rows 0..255 of the slice-major buffer: its row stride is ln_part_rows = M)
        assert torch.equal(y_p, y[:256])
    # ... and the producer's extra outputs do not depend on the kernel either
    x_s = xd[:256].clone()
    xb_s = torch.empty(256, C_, device=DEV, dtype=ht)
    part_s = torch.empty(C_ // 64, 256, 2, device=DEV, dtype=torch.float32)
    o.gemm(aod[:256], wod, x_s, M=256, N=C_, K=C_, lda=C_, ldy=C_, bias=bod, residual=x_s, ldr=C_, y2=xb_s, ldy2=C_, ln_part=part_s)
    assert torch.equal(x_s, x_new[:256]) and torch.equal(xb_s, xb[:256]) and torch.equal(part_s, part[:, :256])
    # loud on shapes the folded forms do not take (the partial-sum form exists in the few-rows kernel only)
    if M > 1024:
        with pytest.raises(Exception):
            o.gemm(xb, wq.to(DEV), y, M=M, N=4 * C_, K=C_, lda=C_, ldy=4 * C_, bias=c.to(DEV), ln_part=part, ln_eps=eps, ln_colsum=s.to(DEV))
    with pytest.raises(Exception):
        o.gemm(xb[:100], wq.to(DEV), y, M=100, N=4 * C_, K=C_, lda=C_, ldy=4 * C_, bias=c.to(DEV), ln_stats=stats, ln_colsum=s.to(DEV))


@pytest.mark.parametrize("ht", [torch.bfloat16, torch.float16])
def test_row_stats_of_bf16_rows(ht):
    """mage_row_stats: (mean, rstd) of bf16 rows = LayerNorm statistics of the rows the frame fill wrote (block 0's ln_1 in bf16 mode),
    consumed by the LayerNorm-folded Linear exactly like mage_ln_stats' output."""
    o = ops()
    for rows, C_ in ((1000, 512), (256, 1024), (77, 64)):
        x = (rnd(rows, C_, seed=rows) * 1.5 + 0.2).to(ht).to(DEV)
        st = torch.empty(rows, 2, device=DEV, dtype=torch.float32)
        o.row_stats(x, 1e-5, st)
        xd = x.double()
        torch.testing.assert_close(st[:, 0].double(), xd.mean(-1), atol=2e-6, rtol=1e-6)
        torch.testing.assert_close(st[:, 1].double(), (xd.var(-1, unbiased=False) + 1e-5).rsqrt(), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("ht", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [512, 2048, 65536])
def test_x_plus_linear_on_a_bf16_residual_stream(M, ht):
    """The producer forms of the 16-bit modes' residual stream (mage_hip.h, ln_part without y2): the residual x is read as 16-bit rows, the new
    rows leave as 16-bit rows only, the LayerNorm partial sums are those of the fp32 values before rounding.  The residual is added in the
    epilogue: y = (A W^T + b) + x in fp32, the reference's own order.  Checked against fp64 on the same operands (one rounding of the fp32
    sum), against the general epilogue's fp32 rows (same sum: equal after rounding), on all three kernels (M = 512 few-rows, M = 2048
    lockstep, M = 65536 8-phase) and the same rows through another kernel (bitwise)."""
    o = ops()
    C_ = 512
    xb0 = (rnd(M, C_, seed=11) + 0.3).to(ht).to(DEV)
    ao = rnd(M, C_, seed=12).to(ht).to(DEV)
    wo, bo = rnd(C_, C_, seed=13, scale=C_ ** -0.5).to(ht).to(DEV), rnd(C_, seed=14, scale=0.1).to(DEV)
    want = xb0.double().cpu() + ao.double().cpu() @ wo.double().cpu().t() + bo.double().cpu()
    # 16-bit residual in, 16-bit rows out, in place, with the partial sums
    xb = xb0.clone()
    part = torch.empty(C_ // 64, M, 2, device=DEV, dtype=torch.float32)
    o.gemm(ao, wo, xb, M=M, N=C_, K=C_, lda=C_, ldy=C_, bias=bo, residual=xb, ldr=C_, ln_part=part)
    # fp32 rows of the same sum from the general epilogue (a 16-bit residual with fp32 rows out): (acc + b) + x
    y32 = torch.empty(M, C_, device=DEV, dtype=torch.float32)
    o.gemm(ao, wo, y32, M=M, N=C_, K=C_, lda=C_, ldy=C_, bias=bo, residual=xb0, ldr=C_)
    torch.testing.assert_close(y32.double().cpu(), want, atol=2e-3, rtol=1e-3)
    assert torch.equal(xb, y32.to(ht))                                  # the stream rows = that fp32 sum, rounded once
    ys = y32.double().view(M, C_ // 64, 64)
    torch.testing.assert_close(part[..., 0].t().double(), ys.sum(-1), atol=2e-4, rtol=1e-5)
    torch.testing.assert_close(part[..., 1].t().double(), (ys * ys).sum(-1), atol=2e-3, rtol=1e-5)
    # out of place, and without the LayerNorm after it (the last block): the same rows
    y16 = torch.empty_like(xb0)
    o.gemm(ao, wo, y16, M=M, N=C_, K=C_, lda=C_, ldy=C_, bias=bo, residual=xb0, ldr=C_)
    assert torch.equal(y16, xb)
    # fp32 residual in, 16-bit rows out (accumulators seeded with x: another order of the same sum)
    xb1 = torch.empty_like(xb0)
    part1 = torch.empty_like(part)
    o.gemm(ao, wo, xb1, M=M, N=C_, K=C_, lda=C_, ldy=C_, bias=bo, residual=xb0.float(), ldr=C_, ln_part=part1)
    torch.testing.assert_close(xb1.float(), xb.float(), **HTOL[ht])
    torch.testing.assert_close(part1, part, atol=2e-3, rtol=1e-4)
    # the same rows through the other kernel: bitwise
    xs = xb0[:256].clone()
    ps = torch.empty(C_ // 64, 256, 2, device=DEV, dtype=torch.float32)
    o.gemm(ao[:256], wo, xs, M=256, N=C_, K=C_, lda=C_, ldy=C_, bias=bo, residual=xs, ldr=C_, ln_part=ps)
    assert torch.equal(xs, xb[:256]) and torch.equal(ps, part[:, :256])
    # ragged row count (edge tiles: clamped residual rows, predicated stores), no partial sums
    Mr = M - 24
    yr = torch.full((M, C_), 7.0, device=DEV, dtype=ht)
    o.gemm(ao, wo, yr, M=Mr, N=C_, K=C_, lda=C_, ldy=C_, bias=bo, residual=xb0, ldr=C_)
    assert torch.equal(yr[:Mr], xb[:Mr]) and (yr[Mr:] == 7.0).all()
    x32 = xb0.float()
    with pytest.raises(Exception):             # fp32 rows out of the partial-sum form need the 16-bit copy y2
        o.gemm(ao, wo, x32, M=M, N=C_, K=C_, lda=C_, ldy=C_, bias=bo, residual=x32, ldr=C_, ln_part=part)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_conv_with_half_resolution_residual_and_narrow_outputs(dt):
    """mage_gemm res_half (the residual of a convolution read at half resolution = nn.Upsample of the skip path folded in) against the
    same convolution over the explicitly upsampled residual, bit for bit; the shapes cover the narrow 256 x 64 tile (N <= 128 with
    enough rows) and the 256-column one."""
    o = ops()
    from mage_amd.modules.vqvae_model import VectorQuantizedVAE as V
    for (N, H, W, cin, cout) in ((2, 8, 8, 16, 32), (3, 16, 16, 64, 256), (70, 32, 32, 64, 64), (70, 32, 32, 64, 128)):
        x = to_dev(rnd(N * H * W, cin, seed=1), dt)
        w = to_dev(rnd(cout, 9 * cin, seed=2, scale=0.05), dt)
        b = rnd(cout, seed=3).to(DEV)
        r_low = to_dev(rnd(N * (H // 2) * (W // 2), cout, seed=4), dt)
        y = torch.empty(N * H * W, cout, device=DEV, dtype=dt)
        V._conv(x, w, y, n_img=N, H=H, W=W, cin=cin, cout=cout, k=3, bias=b, residual=r_low, ldr=cout, post_relu=True, res_half=True)
        r_up = o.upsample2(r_low, torch.empty(N * H * W, cout, device=DEV, dtype=dt), N=N, H=H // 2, W=W // 2, Cc=cout)
        y2 = torch.empty_like(y)
        V._conv(x, w, y2, n_img=N, H=H, W=W, cin=cin, cout=cout, k=3, bias=b, residual=r_up, ldr=cout, post_relu=True)
        assert torch.equal(y, y2)
        # and against torch (fp32 reference of the same op)
        xi = x.float().view(N, H, W, cin).permute(0, 3, 1, 2).cpu()
        wi = w.float().view(cout, 3, 3, cin).permute(0, 3, 1, 2).cpu()
        want = F.relu(F.conv2d(xi, wi, b.cpu(), padding=1) + r_up.float().view(N, H, W, cout).permute(0, 3, 1, 2).cpu())
        torch.testing.assert_close(y.float().view(N, H, W, cout).permute(0, 3, 1, 2).cpu(), want, **TOL[dt])
        # a_half: the convolution's INPUT at half resolution (nn.Upsample in front of it folded into the gather)
        x_low = to_dev(rnd(N * (H // 2) * (W // 2), cin, seed=5), dt)
        x_up = o.upsample2(x_low, torch.empty(N * H * W, cin, device=DEV, dtype=dt), N=N, H=H // 2, W=W // 2, Cc=cin)
        ya, yb = torch.empty_like(y), torch.empty_like(y)
        V._conv(x_low, w, ya, n_img=N, H=H, W=W, cin=cin, cout=cout, k=3, bias=b, act=o.ACT_RELU, a_half=True, a_img_stride=(H // 2) * (W // 2))
        V._conv(x_up, w, yb, n_img=N, H=H, W=W, cin=cin, cout=cout, k=3, bias=b, act=o.ACT_RELU)
        assert torch.equal(ya, yb)
    with pytest.raises(Exception):
        o.gemm(x, w, y, M=N * H * W, N=cout, K=9 * cin, lda=cin, ldy=cout, res_half=True)          # no residual


@pytest.mark.parametrize("M,D,K", [(4099, 1024, 512), (20000, 256, 512), (5000, 32, 64), (777, 128, 1000), (33, 64, 300)])
def test_vq_nearest_on_the_fp64_matrix_cores(M, D, K):
    """mage_vq_nearest (v_mfma_f64_16x16x4_f64 kernel) against the reference FORMULA evaluated with float64 dot products in torch:
    fl32(fl32(|c|^2 + |z|^2) - 2 <z, c>), first minimum: indices and the top-2 margin, for ragged M and K (not multiples of the 32-row /
    64-code blocks), exact codebook hits included."""
    o = ops()
    g = torch.Generator().manual_seed(M + D + K)
    z = torch.randn(M, D, generator=g)
    cb = torch.randn(K, D, generator=g) * 0.8
    n_hit = min(K, M) // 2
    z[:n_hit] = cb[torch.randperm(K, generator=g)[:n_hit]]                 # rows that ARE code vectors
    zd, cbd = z.to(DEV), cb.to(DEV)
    cbt, c2 = o.vq_prepare(cbd)
    ids, mg = o.vq_nearest(zd, cbt, c2, want_margin=True)
    dots = zd.double() @ cbd.double().t()
    s32 = c2[None, :] + (zd.double() ** 2).sum(1).float()[:, None]         # fl32(|c|^2 + |z|^2)
    dist = (s32.double() - 2.0 * dots.float().double()).float()            # fmaf(-2, fl32(dot), s): one rounding
    srt = dist.sort(dim=1, stable=True)                                    # stable: the first minimum wins, as in the reference
    assert torch.equal(ids, srt.indices[:, 0])
    torch.testing.assert_close(mg, srt.values[:, 1] - srt.values[:, 0], atol=0, rtol=0)


@pytest.mark.parametrize("Cd", [128, 512])
def test_table_conv_equals_convolution_of_embeddings_and_folded_linear(Cd):
    """mage_table_conv: conv3x3(embedding(ids)) + positions (+ a Linear folded into the table) as a gather-sum, against torch's conv2d
    on the embedded image (fp64), incl. border pixels, regrouped output rows and the broadcast row table.  Cd = 512 with 16-bit tables and
    rows runs the 8-channels-per-lane kernel (table_conv512_kernel): the same bits as the generic one on the fp32 copy of the table."""
    from mage_amd import ops as o
    g = torch.Generator().manual_seed(8)
    Kc, C, R, B, Lm1, L = 48, 64, 6, 2, 3, 4
    emb = torch.randn(Kc, C, generator=g).to(DEV)
    cw = (torch.randn(C, C, 3, 3, generator=g) * 0.1).to(DEV)                      # [Cout, Cin, kh, kw]
    w_in = (torch.randn(Cd, C, generator=g) * 0.1).to(DEV)
    b_in = torch.randn(Cd, generator=g).to(DEV)
    pos = torch.randn(R * R, C, generator=g).to(DEV)
    tpos = torch.randn(L, Cd, generator=g).to(DEV)
    ids = torch.randint(0, Kc, (B * Lm1, R * R), generator=g).to(DEV)
    T = torch.stack([emb.double() @ cw[:, :, ky, kx].double().t() for ky in range(3) for kx in range(3)]).float().contiguous()     # [9, Kc, C]
    y = o.table_conv(ids, T, torch.empty(B * Lm1 * R * R, C, device=DEV), n_img=B * Lm1, H=R, W=R, pos=pos)
    img = emb[ids].view(B * Lm1, R, R, C).permute(0, 3, 1, 2).double()
    want = torch.nn.functional.conv2d(img, cw.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, C) + pos.double().repeat(B * Lm1, 1)
    torch.testing.assert_close(y.double(), want, rtol=1e-5, atol=1e-5)
    # the decoder form: in_linear folded, frames written into slots 1.. of [B, L, hw, Cd] with the T positions
    T2 = (T.double() @ w_in.double().t()).float().contiguous()
    P2 = (pos.double() @ w_in.double().t() + b_in.double()).float().contiguous()
    x = torch.full((B * L * R * R, Cd), 7.0, device=DEV)
    o.table_conv(ids, T2, x, n_img=B * Lm1, H=R, W=R, pos=P2, rowadd=tpos, rowadd_div=R * R, rowadd_mod=L, group=Lm1 * R * R,
                 y_group_stride=L * R * R, y_off=R * R)
    xw = x.view(B, L, R * R, Cd)
    assert (xw[:, 0] == 7.0).all()                                                  # slot 0 (the motion anchor's) untouched
    want2 = (want @ w_in.double().t() + b_in.double()).view(B, Lm1, R * R, Cd) + tpos.double()[1:].view(1, Lm1, 1, Cd)
    torch.testing.assert_close(xw[:, 1:].double(), want2, rtol=1e-5, atol=2e-5)
    # 16-bit tables and rows (the bf16 / f16 modes' frame fill): entries rounded once, fp32 sums, the row rounded on the way out
    for ht in (torch.bfloat16, torch.float16):
        x16 = torch.full((B * L * R * R, Cd), 7.0, device=DEV, dtype=ht)
        T2h = T2.to(ht)
        o.table_conv(ids, T2h, x16, n_img=B * Lm1, H=R, W=R, pos=P2, rowadd=tpos, rowadd_div=R * R, rowadd_mod=L, group=Lm1 * R * R,
                     y_group_stride=L * R * R, y_off=R * R)
        x32 = torch.full((B * L * R * R, Cd), 7.0, device=DEV)
        o.table_conv(ids, T2h.float().contiguous(), x32, n_img=B * Lm1, H=R, W=R, pos=P2, rowadd=tpos, rowadd_div=R * R, rowadd_mod=L,
                     group=Lm1 * R * R, y_group_stride=L * R * R, y_off=R * R)
        assert torch.equal(x16, x32.to(ht))                                          # same sums, one rounding
    bad = ids.clone()
    bad[0, 0] = Kc
    o.table_conv(bad, T, torch.empty(B * Lm1 * R * R, C, device=DEV), n_img=B * Lm1, H=R, W=R)
    with pytest.raises(ValueError, match="out of range"):
        o.check_device_errors(DEV)


@pytest.mark.parametrize("n_img,H", [(3, 16), (1, 2), (300, 16), (7, 6)])
def test_resblock_table_equals_its_three_launches(n_img, H):
    """mage_resblock_table (the f4 decoder's first ResBlock on codebook rows, vqvae_model.py:111-124,180, in one launch) against the three
    launches it replaces -- mage_embedding into the padded frame buffer, mage_table_conv, the 1x1 mage_gemm with BatchNorm scale / shift,
    the bf16 residual and the trailing ReLU: the same bits (same table-sum order, same MFMA operand placement and k order, same epilogue
    arithmetic), the halo of the padded buffer untouched; and against an fp64 restatement from the embedded frames.  n_img = 300: several
    tiles per workgroup (the double-buffered operand image and the id staging two tiles ahead); H = 2 / 6: every row at an image border."""
    o = ops()
    Kc, C, Wd = 40, 256, 16
    P_w = Wd + 2
    PP = (H + 2) * P_w
    g = torch.Generator().manual_seed(300 + n_img)
    cb = torch.randn(Kc, C, generator=g).to(DEV)
    w3 = (torch.randn(C, C, 3, 3, generator=g) * (9 * C) ** -0.5)                   # [co, ci, ky, kx], BatchNorm folded
    b3 = (torch.randn(C, generator=g) * 0.1).to(DEV)
    w1 = (torch.randn(C, C, generator=g) * C ** -0.5).bfloat16().to(DEV)
    b1, s1, t1 = [(torch.randn(C, generator=g) * sc + off).to(DEV) for sc, off in ((0.1, 0.0), (0.2, 1.0), (0.1, 0.0))]
    ids = torch.randint(0, Kc, (n_img, H, Wd), generator=g).to(DEV)
    rcb = torch.relu(cb)
    T = torch.stack([(rcb.double() @ w3[:, :, ky, kx].double().t().to(DEV)) for ky in range(3) for kx in range(3)]).float().bfloat16().contiguous()
    hw = H * Wd
    for post_relu, scaled in ((True, True), (False, False)):
        # three launches
        pad0 = torch.zeros(n_img * PP + 1, C, device=DEV, dtype=torch.bfloat16)
        o.embedding(ids, cb, pad0, relu=True, group=hw, group_stride=PP, off=P_w + 1, inner=Wd, inner_stride=P_w)
        t = torch.empty(n_img * hw, C, device=DEV, dtype=torch.bfloat16)
        o.table_conv(ids.reshape(-1), T, t, n_img=n_img, H=H, W=Wd, bias=b3, relu=True)
        want = torch.full((n_img * PP + 1, C), 3.0, device=DEV, dtype=torch.bfloat16)
        o.gemm(t, w1, want, M=n_img * hw, N=C, K=C, lda=C, ldy=C, bias=b1, scale=s1 if scaled else None, shift=t1 if scaled else None,
               residual=pad0, ldr=C, post_relu=post_relu, out_h=H, out_w=Wd, y_img_stride=PP, y_mul_y=P_w, y_off=P_w + 1)
        # one launch
        got = torch.full((n_img * PP + 1, C), 3.0, device=DEV, dtype=torch.bfloat16)
        o.resblock_table(ids.reshape(-1), T, cb, w1, got, n_img=n_img, H=H, W=Wd, bias3=b3, b1=b1, scale1=s1 if scaled else None,
                         shift1=t1 if scaled else None, post_relu=post_relu, ldy=C, y_img_stride=PP, y_row_pitch=P_w, y_off=P_w + 1)
        assert torch.equal(got.float().cpu(), want.float().cpu()), (got.float() - want.float()).abs().max().item()
        inner = got[:-1].view(n_img, H + 2, P_w, C)
        assert (inner[:, 0] == 3.0).all() and (inner[:, -1] == 3.0).all() and (inner[:, :, 0] == 3.0).all() and (inner[:, :, -1] == 3.0).all()
        # fp64 from the embedded frames (t rounded to bf16 as the kernel rounds it)
        x = rcb.bfloat16().double()[ids]                                                               # [n, H, W, C]
        xt = torch.zeros(n_img, H + 2, P_w, dtype=torch.long)
        xt[:, 1:-1, 1:-1] = ids.cpu() + 1
        Tz = torch.cat([torch.zeros(9, 1, C, dtype=torch.float64), T.double().cpu()], 1)            # code 0 = outside the image
        tsum = b3.double().cpu() + sum(Tz[ky * 3 + kx][xt[:, ky:ky + H, kx:kx + Wd]] for ky in range(3) for kx in range(3))
        tt = torch.relu(tsum).float().bfloat16().double()
        v = tt @ w1.double().cpu().t() + b1.double().cpu()
        if scaled:
            v = v * s1.double().cpu() + t1.double().cpu()
        v = v + x.cpu()
        if post_relu:
            v = torch.relu(v)
        torch.testing.assert_close(inner[:, 1:-1, 1:-1].double().cpu(), v, atol=3e-2, rtol=2e-2)
    bad = ids.clone()
    bad[0, 0, 0] = Kc
    o.resblock_table(bad.reshape(-1), T, cb, w1, got, n_img=n_img, H=H, W=Wd, bias3=b3, b1=b1, ldy=C, y_img_stride=PP, y_row_pitch=P_w, y_off=P_w + 1)
    with pytest.raises(ValueError, match="out of range"):
        o.check_device_errors(DEV)


@pytest.mark.parametrize("n_img,H,Wd", [(3, 16, 16), (6, 8, 8), (520, 16, 16), (1, 4, 16)])
def test_resblock_rows_equals_the_1x1_gemm(n_img, H, Wd):
    """mage_resblock_rows (the tail of a ResBlock: y = relu(x + BN(conv1x1(t))), t in plain bf16 rows, x and y inside zero-padded frame
    buffers; vqvae_model.py:111-124) against the mage_gemm it replaces (scale / shift, bf16 residual, post_relu, regrouped output rows):
    the same bits, the halo untouched, y allowed to alias x.  n_img = 520: more than two 64-row tiles per workgroup."""
    o = ops()
    C = 256
    P_w = Wd + 2
    PP = (H + 2) * P_w
    hw = H * Wd
    g = torch.Generator().manual_seed(400 + n_img)
    tt = torch.relu(torch.randn(n_img * hw, C, generator=g)).bfloat16().to(DEV)
    xpad = torch.zeros(n_img * PP + 1, C, dtype=torch.bfloat16)
    xpad[:-1].view(n_img, H + 2, P_w, C)[:, 1:-1, 1:-1] = torch.randn(n_img, H, Wd, C, generator=g).bfloat16()
    xpad = xpad.to(DEV)
    w1 = (torch.randn(C, C, generator=g) * C ** -0.5).bfloat16().to(DEV)
    b1, s1, t1 = [(torch.randn(C, generator=g) * sc + off).to(DEV) for sc, off in ((0.1, 0.0), (0.2, 1.0), (0.1, 0.0))]
    for post_relu, scaled in ((True, True), (False, False)):
        kw = dict(scale=s1, shift=t1) if scaled else {}
        want = torch.full((n_img * PP + 1, C), 3.0, device=DEV, dtype=torch.bfloat16)
        o.gemm(tt, w1, want, M=n_img * hw, N=C, K=C, lda=C, ldy=C, bias=b1, residual=xpad, ldr=C, post_relu=post_relu, out_h=H, out_w=Wd,
               y_img_stride=PP, y_mul_y=P_w, y_off=P_w + 1, **kw)
        got = torch.full((n_img * PP + 1, C), 3.0, device=DEV, dtype=torch.bfloat16)
        kw2 = dict(scale1=s1, shift1=t1) if scaled else {}
        o.resblock_rows(tt, w1, xpad, got, n_img=n_img, H=H, W=Wd, b1=b1, post_relu=post_relu, lda=C, ldr=C, ldy=C, img_stride=PP, row_pitch=P_w,
                        off=P_w + 1, **kw2)
        assert torch.equal(got.float().cpu(), want.float().cpu()), (got.float() - want.float()).abs().max().item()
        assert want[:-1].view(n_img, H + 2, P_w, C)[:, 1:-1, 1:-1].float().abs().max().item() > 0.5
        # in place on the frame buffer
        xin = xpad.clone()
        o.resblock_rows(tt, w1, xin, xin, n_img=n_img, H=H, W=Wd, b1=b1, post_relu=post_relu, lda=C, ldr=C, ldy=C, img_stride=PP, row_pitch=P_w,
                        off=P_w + 1, **kw2)
        inner = torch.zeros(n_img * PP + 1, dtype=torch.bool)
        inner[:-1].view(n_img, H + 2, P_w)[:, 1:-1, 1:-1] = True
        assert torch.equal(xin.float().cpu()[inner], want.float().cpu()[inner]) and torch.equal(xin.cpu()[~inner], xpad.cpu()[~inner])
    with pytest.raises(Exception, match="multiple of 64"):
        o.resblock_rows(tt, w1, xpad, got, n_img=1, H=3, W=5, b1=b1, lda=C, ldr=C, ldy=C, img_stride=PP, row_pitch=P_w, off=P_w + 1)


@pytest.mark.parametrize("ht", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,act,ln,f32out", [(65536, 1536, 512, 0, True, False), (32768, 2048, 512, 2, True, False),
                                                  (131072, 512, 512, 0, False, True), (65536, 1024, 1024, 2, False, False),
                                                  (65536, 1024, 256, 0, True, True)])
def test_gemm_one_wave_per_simd_equals_the_8phase_kernel(M, N, K, act, ln, f32out, ht):
    """csrc/gemm4.hip (QKV / c_fc at full-loop sizes: 4 waves of 128x128 outputs, accumulators in literal AGPRs behind inline asm) against the
    8-phase kernel on the same product (library option gemm_no_4w): bit-identical outputs -- same MFMA, same k order, same
    epilogue function -- and both against fp64 on a sample of rows; repeated launches agree (race screen of the hand-placed schedule)."""
    o = ops()
    a = rnd(M, K, seed=11).to(ht)
    w, b = rnd(N, K, seed=12, scale=K ** -0.5).to(ht), rnd(N, seed=13, scale=0.1)
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    kw = dict(M=M, N=N, K=K, lda=K, ldy=N, bias=bd, act=act)
    if ln:
        st = torch.stack([0.05 * rnd(M, seed=14), 1.0 + 0.2 * rnd(M, seed=15).abs()], 1).contiguous()
        cs = 0.3 * rnd(N, seed=16)
        kw.update(ln_stats=st.to(DEV), ln_colsum=cs.to(DEV))
    odt = torch.float32 if f32out else ht
    ys = []
    for no4 in (True, False, False, False):
        with config.lib_option("gemm_no_4w", int(no4)):
            y = torch.full((M, N), float("nan"), device=DEV, dtype=odt)
            o.gemm(ad, wd, y, **kw)
            ys.append(y)
    torch.cuda.synchronize()
    for y in ys[1:]:
        assert torch.equal(y, ys[0])
    rows = torch.arange(0, M, M // 64) + 5
    acc = a[rows].double() @ w.double().t()
    want = ((acc - st[rows, 0:1].double() * cs.double()) * st[rows, 1:2].double() if ln else acc) + b.double()
    if act == o.ACT_QUICKGELU:
        want = want * torch.sigmoid(1.702 * want)
    tol = dict(atol=2e-3, rtol=1e-4) if f32out else HTOL[ht]
    torch.testing.assert_close(ys[1][rows.to(DEV)].cpu().double(), want, **tol)


def test_gemm_one_wave_per_simd_race_screen_under_memory_load():
    """The hand-placed schedule of csrc/gemm4.hip orders its LDS-DMA writes and fragment reads by counted waits + ONE barrier per slab; an early
    read would still pass whenever the DMA happens to land first.  So: 24 launches of the QKV-shaped product while a second stream keeps the HBM
    busy with large copies (the DMA latency the schedule sees changes from launch to launch), every output compared bit for bit with the
    8-phase kernel's."""
    o = ops()
    M, N, K = 65536, 1536, 512
    a = rnd(M, K, seed=21).bfloat16().to(DEV)
    w, b = rnd(N, K, seed=22, scale=K ** -0.5).bfloat16().to(DEV), rnd(N, seed=23, scale=0.1).to(DEV)
    st = torch.stack([0.05 * rnd(M, seed=24), 1.0 + 0.2 * rnd(M, seed=25).abs()], 1).contiguous().to(DEV)
    cs = (0.3 * rnd(N, seed=26)).to(DEV)
    kw = dict(M=M, N=N, K=K, lda=K, ldy=N, bias=b, ln_stats=st, ln_colsum=cs)
    with config.lib_option("gemm_no_4w", 1):
        ref = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        o.gemm(a, w, ref, **kw)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.empty(1 << 28, device=DEV, dtype=torch.uint8)
    big2 = torch.empty_like(big)
    outs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(4)]
    bad = 0
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(rep + 1):                  # a different amount of competing traffic every round
                big2.copy_(big, non_blocking=True)
        for y in outs:
            y.fill_(float("nan"))
            o.gemm(a, w, y, **kw)
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(y, ref)) for y in outs)
    assert bad == 0, f"{bad} of 24 launches differ from the 8-phase kernel's output"


def test_gemm_one_wave_per_simd_training_forms_equal_the_8phase_kernel():
    """The forms the training path sends to csrc/gemm4.hip: no bias (data-gradient GEMMs), c_fc with pre-activation AND activated rows (y2 with
    QuickGELU), the data gradient times QuickGELU'(saved rows) (MAGE_ACT_QUICKGELU_GRAD): bit-identical to the 8-phase kernel's outputs."""
    o = ops()
    M, N, K = 65536, 1024, 512
    a = rnd(M, K, seed=31).bfloat16().to(DEV)
    w, b = rnd(N, K, seed=32, scale=K ** -0.5).bfloat16().to(DEV), rnd(N, seed=33, scale=0.1).to(DEV)
    pre_saved = (1.5 * rnd(M, N, seed=34)).bfloat16().to(DEV)

    def run(form):
        y = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
        y2 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
        if form == "nobias":
            o.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N)
            return (y,)
        if form == "dual":
            o.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N, bias=b, act=o.ACT_QUICKGELU, y2=y2, ldy2=N)
            return (y, y2)
        o.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N, act=o.ACT_QUICKGELU_GRAD, y2=pre_saved, ldy2=N)
        return (y,)
    for form in ("nobias", "dual", "gelu_grad"):
        with config.lib_option("gemm_no_4w", 1):
            ref = run(form)
        with config.lib_option("gemm4_train_forms", 1):   # the two c_fc forms stay on the 8-phase kernel by default (slower here): ask for them
            got = run(form)
        torch.cuda.synchronize()
        for r, g_ in zip(ref, got):
            assert not torch.isnan(g_.float()).any() and torch.equal(r, g_), form


@pytest.mark.parametrize("ht", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,act,ln", [(65536, 2048, 2, True), (65792, 2048, 2, True), (262144, 512, 2, False), (65536, 1536, 0, True),
                                        (33024, 1536, 0, True), (65536, 1024, 0, False),
                                        (16384, 2048, 2, True), (16384, 1536, 0, True), (8192, 2048, 2, True), (8192, 1536, 0, True)])      # the incremental step
def test_gemm_split_half_kernel_equals_gemm4_and_the_8phase_kernel(M, N, act, ln, ht):
    """csrc/gemm4h.hip (round 6: the wave's 128 x 128 block as two 64-row halves, the finished half's epilogue between the other half's MFMAs, rows
    stored straight from the swapped-role accumulator layout) against gemm4_kernel and the 8-phase kernel on the same product: bit-identical
    outputs in bf16 and f16, the LayerNorm-consuming and plain forms, with and without QuickGELU, whole and ragged tile counts per workgroup
    (257 and 129 row tiles: the last pass of some workgroups drains alone); repeated launches agree."""
    o = ops()
    K = 512
    a = rnd(M, K, seed=41).to(ht)
    w, b = rnd(N, K, seed=42, scale=K ** -0.5).to(ht), rnd(N, seed=43, scale=0.1)
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    kw = dict(M=M, N=N, K=K, lda=K, ldy=N, bias=bd, act=act)
    if ln:
        st = torch.stack([0.05 * rnd(M, seed=44), 1.0 + 0.2 * rnd(M, seed=45).abs()], 1).contiguous()
        cs = 0.3 * rnd(N, seed=46)
        kw.update(ln_stats=st.to(DEV), ln_colsum=cs.to(DEV))

    def run(**opts):
        import contextlib
        with contextlib.ExitStack() as es:
            for k_, v_ in opts.items():
                es.enter_context(config.lib_option(k_, v_))
            y = torch.full((M, N), float("nan"), device=DEV, dtype=ht)
            o.gemm(ad, wd, y, **kw)
            return y
    y8 = run(gemm_no_4w=1)
    y4 = run(gemm_no_4h=1)
    yh = [run(gemm_4h_plain=1) for _ in range(3)]
    torch.cuda.synchronize()
    assert not torch.isnan(yh[0].float()).any()
    assert torch.equal(y4, y8)
    for y in yh:
        assert torch.equal(y, y4), f"{int((y != y4).sum())} of {y.numel()} outputs differ from gemm4_kernel's"
    rows = torch.arange(0, M, M // 64) + 5
    acc = a[rows].double() @ w.double().t()
    want = ((acc - st[rows, 0:1].double() * cs.double()) * st[rows, 1:2].double() if ln else acc) + b.double()
    if act == o.ACT_QUICKGELU:
        want = want * torch.sigmoid(1.702 * want)
    torch.testing.assert_close(yh[0][rows.to(DEV)].cpu().double(), want, **HTOL[ht])


def test_gemm_split_half_kernel_race_screen_under_memory_load():
    """gemm4h_kernel orders its LDS-DMA writes, fragment reads, constants and row stores by COUNTED vmcnt waits + one barrier per step (the count
    leaves the A pieces, the constants' pieces and the stores in flight): a wrong count would still pass whenever the pieces happen to land in
    time.  24 launches of the c_fc- and QKV-shaped products while a second stream keeps the HBM busy, every output compared bit for bit with
    gemm4_kernel's."""
    o = ops()
    K = 512
    for M, N, act in ((65536, 2048, o.ACT_QUICKGELU), (65536, 1536, 0)):
        a = rnd(M, K, seed=51).bfloat16().to(DEV)
        w, b = rnd(N, K, seed=52, scale=K ** -0.5).bfloat16().to(DEV), rnd(N, seed=53, scale=0.1).to(DEV)
        st = torch.stack([0.05 * rnd(M, seed=54), 1.0 + 0.2 * rnd(M, seed=55).abs()], 1).contiguous().to(DEV)
        cs = (0.3 * rnd(N, seed=56)).to(DEV)
        kw = dict(M=M, N=N, K=K, lda=K, ldy=N, bias=b, ln_stats=st, ln_colsum=cs, act=act)
        with config.lib_option("gemm_no_4h", 1):
            ref = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            o.gemm(a, w, ref, **kw)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        big = torch.empty(1 << 28, device=DEV, dtype=torch.uint8)
        big2 = torch.empty_like(big)
        outs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(4)]
        bad = 0
        with config.lib_option("gemm_4h_plain", 1):
            for rep in range(6):
                with torch.cuda.stream(side):
                    for _ in range(rep + 1):
                        big2.copy_(big, non_blocking=True)
                for y in outs:
                    y.fill_(float("nan"))
                    o.gemm(a, w, y, **kw)
                torch.cuda.synchronize()
                bad += sum(int(not torch.equal(y, ref)) for y in outs)
        assert bad == 0, f"{bad} of 24 launches (N = {N}) differ from gemm4_kernel's output"


@pytest.mark.parametrize("M,N,act", [(65536, 2048, 2), (16384, 1536, 0)])
def test_gemm_split_half_kernel_row_offsets(M, N, act):
    """a_off / y_off (rows of A skipped, rows of Y skipped: plain rows otherwise) go through gemm4h_kernel's base pointers: same bits as the 8-phase
    kernel, nothing written outside the addressed rows."""
    o = ops()
    K, a_off, y_off = 512, 37, 53
    a = rnd(M + a_off + 8, K, seed=61).bfloat16().to(DEV)
    w, b = rnd(N, K, seed=62, scale=K ** -0.5).bfloat16().to(DEV), rnd(N, seed=63, scale=0.1).to(DEV)
    st = torch.stack([0.05 * rnd(M, seed=64), 1.0 + 0.2 * rnd(M, seed=65).abs()], 1).contiguous().to(DEV)
    cs = (0.3 * rnd(N, seed=66)).to(DEV)
    kw = dict(M=M, N=N, K=K, lda=K, ldy=N, bias=b, ln_stats=st, ln_colsum=cs, act=act, a_off=a_off, y_off=y_off, a_img_stride=M, y_img_stride=M)
    outs = []
    for opt in (dict(gemm_no_4w=1), dict(gemm_4h_plain=1)):
        import contextlib
        with contextlib.ExitStack() as es:
            for k_, v_ in opt.items():
                es.enter_context(config.lib_option(k_, v_))
            y = torch.full((M + y_off + 8, N), 7.0, device=DEV, dtype=torch.bfloat16)
            o.gemm(a, w, y, **kw)
            outs.append(y)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert (outs[1][:y_off] == 7.0).all() and (outs[1][M + y_off:] == 7.0).all() and not (outs[1][y_off:M + y_off] == 7.0).all()
