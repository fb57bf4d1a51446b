"""Training path on the GPU: every backward kernel against PyTorch autograd of the same op, the whole
``loss.backward()`` of MAGE.forward against the CPU oracle's autograd (the oracle is a differentiable restatement of the
reference's forward, pinned to the reference's goldens), and the fused Adam step against torch.optim.Adam."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mage_amd import config
from mage_amd.utils import synth
from oracle import mage_oracle as O
from tests.helpers import build_mage, cpu_sd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRAD_TOL = 1e-4          # relative to the largest entry of the reference gradient tensor


def ops():
    from mage_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def rel(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item() / max(b.detach().abs().max().item(), 1e-30)


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_transpose_plain_and_conv_taps(dtype):
    o = ops()
    M, Cc = 1000, 72
    x = rnd(M + 5, Cc, seed=1).to(dtype).to(DEV)
    Mp = 1024
    y = torch.full((Cc, Mp), 7.0, device=DEV, dtype=dtype)
    o.transpose(x, y, M=M, Mp=Mp, C=Cc, ldx=Cc, ldy=Mp)
    assert torch.equal(y[:, :M], x[:M].t()) and (y[:, M:] == 0).all()
    # the x[:, 1:] rows of a [B, L, hw] stream
    B, L, hw = 3, 4, 16
    s = rnd(B * L * hw, 8, seed=2).to(dtype).to(DEV)
    M1 = B * (L - 1) * hw
    y = torch.empty(8, 192, device=DEV, dtype=dtype)
    o.transpose(s, y, M=M1, Mp=192, C=8, ldx=8, ldy=192, out_w=(L - 1) * hw, img_stride=L * hw, a_off=hw)
    assert torch.equal(y[:, :M1], s.view(B, L, hw, 8)[:, 1:].reshape(M1, 8).t())
    # conv taps: y[tap*C + c, m] = x[shifted pixel, c] with zero padding
    n, R = 2, 4
    img = rnd(n * R * R, 8, seed=3).to(dtype).to(DEV)
    y = torch.empty(9 * 8, 64, device=DEV, dtype=dtype)
    for ky in range(3):
        for kx in range(3):
            o.transpose(img, y, M=n * R * R, Mp=64, C=8, ldx=8, ldy=64, y_row0=(ky * 3 + kx) * 8, out_h=R, out_w=R, in_h=R, in_w=R,
                        img_stride=R * R, dy=ky - 1, dx=kx - 1)
    pad = F.pad(img.float().view(n, R, R, 8), (0, 0, 1, 1, 1, 1))
    for ky in range(3):
        for kx in range(3):
            want = pad[:, ky:ky + R, kx:kx + R].reshape(n * R * R, 8).t()
            assert torch.equal(y[(ky * 3 + kx) * 8:(ky * 3 + kx + 1) * 8, :n * R * R].float(), want)


@pytest.mark.parametrize("dtype,M,N,K", [(torch.float32, 5000, 64, 192), (torch.bfloat16, 70000, 512, 256), (torch.float32, 704, 1536, 512),
                                         (torch.bfloat16, 40000, 2048, 512)])
def test_weight_gradient_split_k(dtype, M, N, K):
    """dW = dY^T X and db = column sums through transposes + ONE split-K launch + fixed-order partial sums."""
    from mage_amd.modules.mage_train import _wgrad
    dy = rnd(M, N, seed=4, scale=0.1).to(dtype)
    x = rnd(M, K, seed=5).to(dtype)
    dW, db = _wgrad(dy.to(DEV), x.to(DEV), M=M, N=N, K=K, ld_dy=N, ld_x=K)
    want = dy.double().t() @ x.double()
    tol = 2e-6 if dtype == torch.float32 else 2e-5          # fp32 accumulation of exact products (bf16 inputs are exact in fp32)
    assert rel(dW.double(), want) < tol * np.sqrt(M)
    assert rel(db.double(), dy.double().sum(0)) < tol * np.sqrt(M)
    dW2, _ = _wgrad(dy.to(DEV), x.to(DEV), M=M, N=N, K=K, ld_dy=N, ld_x=K)
    assert torch.equal(dW, dW2)                              # deterministic


@pytest.mark.parametrize("Cc,rows,dy_dtype", [(512, 4099, torch.float32), (64, 300, torch.float32), (512, 2048, torch.bfloat16), (1024, 77, torch.float32)])
def test_layernorm_backward(Cc, rows, dy_dtype):
    o = ops()
    x = (rnd(rows, Cc, seed=6) * 2 + 0.3).requires_grad_()
    g, b = (1 + 0.1 * rnd(Cc, seed=7)).requires_grad_(), rnd(Cc, seed=8).requires_grad_()
    dy = rnd(rows, Cc, seed=9).to(dy_dtype)
    F.layer_norm(x, (Cc,), g, b, 1e-5).backward(dy.float())
    dx0 = rnd(rows, Cc, seed=10)
    dx = dx0.to(DEV).clone()
    dg, db = o.layernorm_bwd(x.detach().to(DEV), g.detach().to(DEV), dy.to(DEV), dx, eps=1e-5, accumulate=True)
    assert rel(dx.cpu() - dx0, x.grad) < 2e-5 and rel(dg, g.grad) < 2e-5 and rel(db, b.grad) < 2e-5
    dx2 = torch.empty(rows, Cc, device=DEV)
    o.layernorm_bwd(x.detach().to(DEV), g.detach().to(DEV), dy.to(DEV), dx2, eps=1e-5, accumulate=False)
    assert rel(dx2, x.grad) < 2e-5


@pytest.mark.parametrize("p,seed", [(0.0, 0), (0.1, 4321)])
def test_layernorm_backward_also_writes_the_next_branch_operand(p, seed):
    """mage_layernorm_bwd's dx_bf16: the updated dx rows through mage_dropout's mask as bf16 (p = 0: the plain cast), bit for bit what the
    separate mage_dropout / cast pass over the updated dx writes; dx itself and (dgamma, dbeta) are unchanged by the extra output."""
    o = ops()
    rows, Cc = 1027, 512
    x, g = (rnd(rows, Cc, seed=6) * 2 + 0.3).to(DEV), (1 + 0.1 * rnd(Cc, seed=7)).to(DEV)
    dy = rnd(rows, Cc, seed=9).to(DEV).to(torch.bfloat16)
    dx0 = rnd(rows, Cc, seed=10).to(DEV)
    dx_a, dx_b = dx0.clone(), dx0.clone()
    ga, ba = o.layernorm_bwd(x, g, dy, dx_a, eps=1e-5, accumulate=True)
    dxb = torch.full((rows, Cc), 7.0, device=DEV, dtype=torch.bfloat16)
    gb, bb = o.layernorm_bwd(x, g, dy, dx_b, eps=1e-5, accumulate=True, dx_bf16=dxb, p=p, seed=seed)
    assert torch.equal(dx_a, dx_b) and torch.equal(ga, gb) and torch.equal(ba, bb)
    want = o.dropout(dx_a, torch.empty(rows, Cc, device=DEV, dtype=torch.bfloat16), p, seed) if p > 0 else dx_a.to(torch.bfloat16)
    assert torch.equal(dxb, want)


@pytest.mark.parametrize("x_dtype,yn_dtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16)])
def test_dropout_add_layernorm_is_the_two_passes_in_one(x_dtype, yn_dtype):
    """mage_dropout_add_layernorm: y = r + dropout(x) equals mage_dropout_add's rows bit for bit; yn = mage_layernorm of those rows."""
    o = ops()
    rows, Cc, p, seed = 515, 512, 0.1, 987
    x, r = rnd(rows, Cc, seed=1).to(DEV).to(x_dtype), rnd(rows, Cc, seed=2).to(DEV)
    g, b = (1 + 0.1 * rnd(Cc, seed=3)).to(DEV), rnd(Cc, seed=4).to(DEV)
    y0 = o.dropout_add(x, r, torch.empty_like(r), p, seed)
    yn0 = o.layernorm(y0, g, b, torch.empty(rows, Cc, device=DEV, dtype=yn_dtype), 1e-5)
    y, yn = o.dropout_add_layernorm(x, r, torch.empty_like(r), g, b, torch.empty(rows, Cc, device=DEV, dtype=yn_dtype), 1e-5, p, seed)
    assert torch.equal(y, y0)
    tol = 2e-6 if yn_dtype == torch.float32 else 1.6e-2             # one bf16 ulp where the two kernels' fp32 values straddle a rounding point
    assert (yn.float() - yn0.float()).abs().max().item() <= tol * max(1.0, yn0.float().abs().max().item())


@pytest.mark.parametrize("kind", ["quickgelu", "gelu", "relu"])
def test_activation_forward_backward(kind):
    o = ops()
    code = {"quickgelu": o.ACT_QUICKGELU, "gelu": o.ACT_GELU_ERF, "relu": o.ACT_RELU}[kind]
    x = (rnd(333, 64, seed=11) * 3).requires_grad_()
    y = {"quickgelu": lambda t: t * torch.sigmoid(1.702 * t), "gelu": F.gelu, "relu": F.relu}[kind](x)
    dy = rnd(333, 64, seed=12)
    y.backward(dy)
    got = o.act(x.detach().to(DEV), torch.empty(333, 64, device=DEV), code)
    assert rel(got, y) < 2e-6
    dx = o.act_bwd(x.detach().to(DEV), dy.to(DEV), torch.empty(333, 64, device=DEV), code)
    assert rel(dx, x.grad) < 5e-6
    xb = x.detach().bfloat16()
    gb = o.act(xb.to(DEV), torch.empty(333, 64, device=DEV, dtype=torch.bfloat16), code)
    assert rel(gb.float(), y) < 1e-2


def test_cross_entropy_backward_and_embedding_scatter():
    o = ops()
    rows, K = 777, 64
    lg = (rnd(rows, K, seed=13) * 2).requires_grad_()
    tg = torch.randint(0, K, (rows,), generator=torch.Generator().manual_seed(14))
    loss = F.cross_entropy(lg, tg)
    (loss * 0.7).backward()
    got = o.cross_entropy_bwd(lg.detach().to(DEV), tg.to(DEV), torch.tensor([0.7], device=DEV), torch.empty(rows, K, device=DEV))
    assert rel(got, lg.grad) < 2e-6
    # embedding scatter with a padding row and the grouped output addressing
    V, Cc, n = 30, 64, 500
    ids = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(15))
    dout = rnd(n, Cc, seed=16)
    want = torch.zeros(V, Cc).index_add_(0, ids[ids != 0], dout[ids != 0])
    got = o.embedding_bwd(ids.to(DEV), dout.to(DEV), torch.zeros(V, Cc, device=DEV), padding_idx=0)
    assert rel(got, want) < 1e-5


def test_group_rowsum_tables():
    o = ops()
    B, L, hw, Cc = 3, 5, 16, 64
    x = rnd(B * L * hw, Cc, seed=17)
    got = o.group_rowsum(x.to(DEV), torch.empty(L, Cc, device=DEV), rows=B * L * hw, C=Cc, div=hw, mod=L)
    assert rel(got, x.view(B, L, hw, Cc).sum((0, 2))) < 1e-5
    got = o.group_rowsum(x.to(DEV), torch.empty(hw, Cc, device=DEV), rows=B * L * hw, C=Cc, div=1, mod=hw)
    assert rel(got, x.view(B * L, hw, Cc).sum(0)) < 1e-5
    sp = torch.rand(B * L)
    got = o.group_rowsum(x.to(DEV), torch.empty(1, Cc, device=DEV), rows=B * L * hw, C=Cc, div=1, mod=1, row_scale=sp.to(DEV), row_scale_div=hw)
    assert rel(got, (x.view(B * L, hw, Cc) * sp[:, None, None]).sum((0, 1))[None]) < 1e-5


def _sdpa_ref(q, k, v, mask):
    s = (q @ k.transpose(-1, -2)) * 32 ** -0.5
    s = s.masked_fill(~mask, float("-inf"))
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("axis,dtype", [(0, torch.float32), (1, torch.float32), (2, torch.float32), (0, torch.bfloat16), ("text", torch.float32),
                                        ("cross", torch.float32)])
def test_attention_backward(axis, dtype):
    """dq, dk, dv of the strided short-sequence attention for every geometry the model uses: axial L (causal) / H / W over a
    packed qkv, text self-attention with key padding, motion-anchor cross-attention (256 queries over S keys)."""
    o = ops()
    H = 2
    Cc = H * 32
    if axis in (0, 1, 2):
        B, L, hh, ww = 2, 5, 4, 4
        hw, M = hh * ww, B * L * hh * ww
        qkv = rnd(M, 3 * Cc, seed=20).to(dtype)
        do = rnd(M, Cc, seed=21).to(dtype)
        geo = [dict(n_seq=B * hw, inner=hw, nq=L, nk=L, q_outer_stride=L * hw, q_axis_stride=hw, causal=True),
               dict(n_seq=B * L * ww, inner=ww, nq=hh, nk=hh, q_outer_stride=hw, q_axis_stride=ww, causal=False),
               dict(n_seq=B * L * hh, inner=1, nq=ww, nk=ww, q_outer_stride=ww, q_axis_stride=1, causal=False)][axis]
        geo.update(kv_outer_stride=geo["q_outer_stride"], kv_axis_stride=geo["q_axis_stride"], n_head=H)
        x = qkv.float().view(B, L, hh, ww, 3, H, 32).requires_grad_()
        perm = [(0, 2, 3, 4, 1, 5), (0, 1, 3, 4, 2, 5), (0, 1, 2, 4, 3, 5)][axis]           # -> [..., head, axis, 32]
        q, k, v = (x[:, :, :, :, j].permute(*perm) for j in range(3))
        n = q.shape[-2]
        mask = torch.ones(n, n, dtype=torch.bool).tril() if axis == 0 else torch.ones(n, n, dtype=torch.bool)
        out = _sdpa_ref(q, k, v, mask)
        inv = np.argsort(perm).tolist()
        out.permute(*inv).reshape(M, Cc).backward(do.float())
        want = x.grad.reshape(M, 3 * Cc)
        dqkv = torch.empty(M, 3 * Cc, device=DEV, dtype=dtype)
        dq = qkv.to(DEV)
        o.attention_bwd(dq, dq[:, Cc:], dq[:, 2 * Cc:], do.to(DEV), dqkv, dqkv[:, Cc:], dqkv[:, 2 * Cc:], ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc,
                        ldo=Cc, ld_dq=3 * Cc, ld_dk=3 * Cc, ld_dv=3 * Cc, **geo)
        assert rel(dqkv, want) < (1e-5 if dtype == torch.float32 else 2e-2)
        return
    B, S = 3, 12
    if axis == "text":
        lens = torch.tensor([12, 7, 9])
        qkv = rnd(B * S, 3 * Cc, seed=22)
        do = rnd(B * S, Cc, seed=23)
        x = qkv.view(B, S, 3, H, 32).requires_grad_()
        q, k, v = (x[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        mask = (torch.arange(S)[None, :] < lens[:, None])[:, None, None, :].expand(B, H, S, S)
        _sdpa_ref(q, k, v, mask).permute(0, 2, 1, 3).reshape(B * S, Cc).backward(do)
        dqkv = torch.empty(B * S, 3 * Cc, device=DEV)
        dq = qkv.to(DEV)
        o.attention_bwd(dq, dq[:, Cc:], dq[:, 2 * Cc:], do.to(DEV), dqkv, dqkv[:, Cc:], dqkv[:, 2 * Cc:], ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc,
                        ldo=Cc, ld_dq=3 * Cc, ld_dk=3 * Cc, ld_dv=3 * Cc, n_seq=B, inner=1, nq=S, nk=S, n_head=H, q_outer_stride=S,
                        q_axis_stride=1, kv_outer_stride=S, kv_axis_stride=1, kv_len=lens.to(torch.int32).to(DEV), kv_len_div=1)
        assert rel(dqkv, x.grad.reshape(B * S, 3 * Cc)) < 1e-5
        return
    nq = 256                                                                      # cross: 256 queries per clip over S text keys
    qp = rnd(B * nq, Cc, seed=24).requires_grad_()
    kvp = rnd(B * S, 2 * Cc, seed=25).requires_grad_()
    do = rnd(B * nq, Cc, seed=26)
    q = qp.view(B, nq, H, 32).permute(0, 2, 1, 3)
    k, v = (kvp.view(B, S, 2, H, 32)[:, :, j].permute(0, 2, 1, 3) for j in range(2))
    _sdpa_ref(q, k, v, torch.ones(nq, S, dtype=torch.bool)).permute(0, 2, 1, 3).reshape(B * nq, Cc).backward(do)
    dqp, dkvp = torch.empty(B * nq, Cc, device=DEV), torch.empty(B * S, 2 * Cc, device=DEV)
    dkv = kvp.detach().to(DEV)
    o.attention_bwd(qp.detach().to(DEV), dkv, dkv[:, Cc:], do.to(DEV), dqp, dkvp, dkvp[:, Cc:], ldq=Cc, ldk=2 * Cc, ldv=2 * Cc, ldo=Cc, ld_dq=Cc,
                    ld_dk=2 * Cc, ld_dv=2 * Cc, n_seq=B, inner=1, nq=nq, nk=S, n_head=H, q_outer_stride=nq, q_axis_stride=1, kv_outer_stride=S,
                    kv_axis_stride=1)
    assert rel(dqp, qp.grad) < 1e-5 and rel(dkvp, kvp.grad) < 1e-5


@pytest.mark.parametrize("axis,L,hh,ww,H", [(0, 16, 4, 4, 6), (0, 20, 2, 3, 4), (0, 5, 4, 4, 2), (1, 3, 16, 4, 5), (2, 2, 4, 16, 16), (1, 2, 32, 2, 4)])
def test_attention_backward_on_the_matrix_cores(axis, L, hh, ww, H):
    """bf16 axial attention backward (attention_bwd_mfma_kernel): 1-2 blocks of 16 queries / keys, causal and full, head counts that do
    not fill the last workgroup, against autograd of the fp32 formula on the same bf16 inputs (the outputs are bf16: 1e-2 of the largest
    entry) and against the thread-per-query kernel."""
    import os
    o = ops()
    Cc, B = H * 32, 2
    hw, M = hh * ww, B * L * hh * ww
    qkv = rnd(M, 3 * Cc, seed=30).bfloat16()
    do = rnd(M, Cc, seed=31).bfloat16()
    geo = [dict(n_seq=B * hw, inner=hw, nq=L, nk=L, q_outer_stride=L * hw, q_axis_stride=hw, causal=True),
           dict(n_seq=B * L * ww, inner=ww, nq=hh, nk=hh, q_outer_stride=hw, q_axis_stride=ww, causal=False),
           dict(n_seq=B * L * hh, inner=1, nq=ww, nk=ww, q_outer_stride=ww, q_axis_stride=1, causal=False)][axis]
    geo.update(kv_outer_stride=geo["q_outer_stride"], kv_axis_stride=geo["q_axis_stride"], n_head=H)
    x = qkv.float().view(B, L, hh, ww, 3, H, 32).requires_grad_()
    perm = [(0, 2, 3, 4, 1, 5), (0, 1, 3, 4, 2, 5), (0, 1, 2, 4, 3, 5)][axis]
    q, k, v = (x[:, :, :, :, j].permute(*perm) for j in range(3))
    n = q.shape[-2]
    mask = torch.ones(n, n, dtype=torch.bool).tril() if axis == 0 else torch.ones(n, n, dtype=torch.bool)
    _sdpa_ref(q, k, v, mask).permute(*np.argsort(perm).tolist()).reshape(M, Cc).backward(do.float())
    want = x.grad.reshape(M, 3 * Cc)

    def run():
        dqkv = torch.full((M, 3 * Cc), float("nan"), device=DEV, dtype=torch.bfloat16)
        dq = qkv.to(DEV)
        o.attention_bwd(dq, dq[:, Cc:], dq[:, 2 * Cc:], do.to(DEV), dqkv, dqkv[:, Cc:], dqkv[:, 2 * Cc:], ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc,
                        ldo=Cc, ld_dq=3 * Cc, ld_dk=3 * Cc, ld_dv=3 * Cc, **geo)
        return dqkv
    got = run()
    assert torch.isfinite(got.float()).all()
    for j, nm in enumerate(("dq", "dk", "dv")):
        assert rel(got[:, j * Cc:(j + 1) * Cc], want[:, j * Cc:(j + 1) * Cc]) < 1e-2, nm
    with config.lib_option("attn_no_mfma", 1):
        old = run()
    assert rel(got, old.float()) < 1e-2


def test_transpose_with_fused_column_sums_and_lds_embedding_scatter():
    """mage_transpose_colsum (the transposed bf16 copy of dY and db = its column sums in one pass, plain and regrouped rows) and the
    LDS-table form of mage_embedding_bwd (small tables, many rows)."""
    o = ops()
    for M, Cc, geo in ((70000, 520, {}), (3 * 5 * 64, 64, dict(out_w=5 * 64, img_stride=6 * 64, a_off=64))):
        rows_src = M if not geo else 3 * 6 * 64
        x = rnd(rows_src, Cc, seed=5).bfloat16()
        Mp = (M + 63) // 64 * 64 + 64
        y = torch.full((Cc, Mp), 7.0, device=DEV, dtype=torch.bfloat16)
        db = o.transpose_colsum(x.to(DEV), y, M=M, Mp=Mp, C=Cc, ldx=Cc, ldy=Mp, **geo)
        src = x.float() if not geo else x.float().view(3, 6 * 64, Cc)[:, 64:].reshape(M, Cc)
        assert torch.equal(y[:, :M].float().cpu(), src.t()) and (y[:, M:] == 0).all()
        torch.testing.assert_close(db.cpu(), src.sum(0), atol=1e-2, rtol=1e-4)
    n, K, Cc = 40000, 512, 128
    ids = torch.randint(0, K, (n,), generator=torch.Generator().manual_seed(6))
    ids[::7] = 3                                                   # a hot row
    g = rnd(n, Cc, seed=7)
    for dt in (torch.float32, torch.bfloat16):
        tab = torch.zeros(K, Cc, device=DEV)
        o.embedding_bwd(ids.to(DEV), g.to(DEV).to(dt), tab)
        want = torch.zeros(K, Cc).index_add_(0, ids, g.to(dt).float())
        torch.testing.assert_close(tab.cpu(), want, atol=2e-3, rtol=1e-4)
        # deterministic: a fixed-order fp32 sum -- rows ascending inside a chunk, chunks ascending (mage_hip.h).  Bit-identical across repeated
        # launches, and equal to that very sum computed on the CPU for a few codes
        for _ in range(3):
            tab2 = torch.zeros(K, Cc, device=DEV)
            o.embedding_bwd(ids.to(DEV), g.to(DEV).to(dt), tab2)
            assert torch.equal(tab2, tab)
        n_chunk = min(64, (n + 4095) // 4096)
        rpc = ((n + n_chunk - 1) // n_chunk + 7) // 8 * 8
        gf = g.to(dt).float()
        for code_ in (3, 0, 511):
            tot = torch.zeros(Cc)
            for c in range(n_chunk):
                part = torch.zeros(Cc)
                for i in (ids[c * rpc:(c + 1) * rpc] == code_).nonzero().flatten().tolist():
                    part = part + gf[c * rpc + i]
                tot = tot + part
            assert torch.equal(tab[code_].cpu(), tot), code_
    # few rows (the caption vocabulary: 30 x 512 table, 704 token rows) take the same deterministic form
    ids_s = torch.randint(0, 30, (704,), generator=torch.Generator().manual_seed(8))
    g_s = rnd(704, 512, seed=9)
    t1 = o.embedding_bwd(ids_s.to(DEV), g_s.to(DEV), torch.zeros(30, 512, device=DEV), padding_idx=0)
    t2 = o.embedding_bwd(ids_s.to(DEV), g_s.to(DEV), torch.zeros(30, 512, device=DEV), padding_idx=0)
    assert torch.equal(t1, t2)
    torch.testing.assert_close(t1.cpu(), torch.zeros(30, 512).index_add_(0, ids_s[ids_s != 0], g_s[ids_s != 0]), atol=1e-4, rtol=1e-5)


def test_dropout_mask_is_stateless_and_scaled():
    o = ops()
    x = torch.ones(1 << 20, device=DEV)
    y1 = o.dropout(x, torch.empty_like(x), 0.1, 1234)
    y2 = o.dropout(x, torch.empty_like(x), 0.1, 1234)
    y3 = o.dropout(x, torch.empty_like(x), 0.1, 1235)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    keep = (y1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 2e-3 and torch.allclose(y1[y1 != 0], torch.tensor(1 / 0.9, device=DEV))
    acc = o.dropout(x, torch.full_like(x, 2.0), 0.1, 1234, accumulate=True)
    assert torch.equal(acc, y1 + 2.0)
    yb = o.dropout(x, torch.empty(1 << 20, device=DEV, dtype=torch.bfloat16), 0.1, 1234)       # the backward's cast + mask in one pass
    assert torch.equal(yb != 0, y1 != 0)


def test_fused_adam_matches_torch():
    o = ops()
    n = 100003
    p0, gs = rnd(n, seed=30), [rnd(n, seed=31 + i, scale=0.1) for i in range(4)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
    p, m, v = p0.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for i, g in enumerate(gs):
        ref.grad = g.clone()
        opt.step()
        o.adam(p, g.to(DEV), m, v, lr=1e-3, beta1=0.9, beta2=0.98, eps=1e-6, step=i + 1)
    assert (p.cpu() - ref.detach()).abs().max().item() < 2e-6


# ------------------------------------------------------------------------------------------------ the whole backward pass
def oracle_grads(sd, batch, L):
    """Gradients of the reference's loss (CPU oracle forward = the reference's forward, autograd of PyTorch CPU)."""
    sd = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.startswith("first_stage_model.") else v) for k, v in sd.items()}
    loss, _ = O.mage_forward_loss(sd, batch, L)
    names = [k for k, v in sd.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    return loss.item(), {k: g for k, g in zip(names, gs)}


@pytest.mark.parametrize("cfg_kw,B,L,seed,batch_kw", [
    (dict(width=64, layers=3, vq_dim=32, K=64), 3, 5, 31, dict(text_len=9, ragged_text=True)),       # the mage_small_d64 fixture's model
    (dict(), 1, 4, 33, dict(digits=2, caption_lengths=(16, 18, 20))),                                # full width (d=512, 6 blocks)
])
def test_loss_backward_matches_oracle_autograd(cfg_kw, B, L, seed, batch_kw):
    """loss.backward() on the HIP path (fp32 mode, eval = no dropout) against autograd through the oracle: every trainable
    parameter's gradient within 1e-4 of the tensor's largest reference entry; unused parameters (ln_q / ln_kv) get zeros."""
    cfg = synth.mnist_model_config(frames_length=L, **cfg_kw)
    m = build_mage(cfg, seed, DEV)
    batch = synth.synth_batch_mnist(B, L, seed=seed, **batch_kw)
    want_loss, want = oracle_grads(cpu_sd(m), batch, L)
    loss, ld = m({k: v.to(DEV) for k, v in batch.items()})
    assert abs(loss.item() - want_loss) < 1e-4 and loss.requires_grad
    loss.backward()
    worst = ("", 0.0)
    n_checked = 0
    for name, p in m.named_parameters():
        if name.startswith("first_stage_model."):
            assert p.grad is None
            continue
        g_ref = want.get(name)
        assert p.grad is not None, name
        if g_ref is None or g_ref.abs().max().item() == 0.0:
            assert p.grad.abs().max().item() == 0.0, name                       # unused by the reference's forward
            continue
        r = rel(p.grad, g_ref)
        n_checked += 1
        if r > worst[1]:
            worst = (name, r)
    print(f"{n_checked} gradients checked, worst relative error {worst[1]:.2e} at {worst[0]}")
    assert worst[1] < GRAD_TOL, worst
    assert n_checked >= 90


def test_c_fc_with_pre_activation_and_activated_rows():
    """mage_gemm LN_DUAL (bf16, act = QuickGELU, y2): one launch writes the pre-activation rows AND QuickGELU of them; against the two-launch
    form (GEMM, then mage_act on its bf16 rows) on both tiled kernels."""
    from mage_amd import ops as o
    g = torch.Generator().manual_seed(3)
    for M in (2048, 65536):
        N, K = 1024, 256
        a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().to(DEV)
        b = (torch.randn(N, generator=g) * 0.1).to(DEV)
        pre_ref = o.gemm(a, w, torch.empty(M, N, device=DEV, dtype=torch.bfloat16), M=M, N=N, K=K, lda=K, ldy=N, bias=b)
        pre = torch.empty_like(pre_ref)
        act = torch.empty_like(pre_ref)
        o.gemm(a, w, pre, M=M, N=N, K=K, lda=K, ldy=N, bias=b, act=o.ACT_QUICKGELU, y2=act, ldy2=N)
        assert torch.equal(pre, pre_ref)
        x = (a.double() @ w.double().t() + b.double()).cpu()
        want = x * torch.sigmoid(1.702 * x)
        torch.testing.assert_close(act.double().cpu(), want, atol=3e-2, rtol=2e-2)


def test_data_gradient_gemm_times_quickgelu_derivative():
    """mage_gemm MAGE_ACT_QUICKGELU_GRAD: y = (a @ w^T) * QuickGELU'(pre) in one launch, against the GEMM followed by mage_act_bwd."""
    from mage_amd import ops as o
    g = torch.Generator().manual_seed(4)
    for M in (2048, 65536):
        N, K = 1024, 256
        a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().to(DEV)
        pre = (torch.randn(M, N, generator=g) * 1.5).bfloat16().to(DEV)
        y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        o.gemm(a, w, y, M=M, N=N, K=K, lda=K, ldy=N, act=o.ACT_QUICKGELU_GRAD, y2=pre, ldy2=N)
        x = pre.double().cpu()
        sg = torch.sigmoid(1.702 * x)
        want = (a.double().cpu() @ w.double().cpu().t()) * (sg * (1 + 1.702 * x * (1 - sg)))
        torch.testing.assert_close(y.double().cpu(), want, atol=3e-2, rtol=2e-2)


def test_bf16_training_gradients_track_fp32():
    cfg = synth.mnist_model_config(frames_length=4, width=64, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, 35, DEV)
    batch = {k: v.to(DEV) for k, v in synth.synth_batch_mnist(2, 4, seed=35).items()}
    m(batch)[0].backward()
    g32 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    m.set_precision("bf16")
    loss, _ = m(batch)
    loss.backward()
    cos = []
    for n, p in m.named_parameters():
        if p.grad is not None and g32[n].abs().max() > 0:
            cos.append(F.cosine_similarity(p.grad.flatten(), g32[n].flatten(), dim=0).item())
    print(f"bf16 vs fp32 gradients: min cosine {min(cos):.4f}, mean {np.mean(cos):.4f}")
    assert min(cos) > 0.98


def test_bf16_training_gradients_track_fp32_with_the_randomness_branch():
    """bf16 training of the CATER family: the Conv3d video prior runs on bf16 operands (fp32 accumulation, statistics and gradient
    streams); its gradients, and everything downstream of it, track the fp32 mode's."""
    L, B, seed = 9, 2, 42
    cfg = synth.cater_model_config(frames_length=L, width=64, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, seed, DEV)
    batch = {k: v.to(DEV) for k, v in synth.synth_batch_cater(B, L, seed=seed, text_len=9).items()}
    batch["reparam_noise"] = torch.randn(B, 64, 16, 16, generator=torch.Generator().manual_seed(seed)).to(DEV)
    l32, _ = m(batch)
    l32.backward()
    g32 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    m.set_precision("bf16")
    l16, _ = m(batch)
    l16.backward()
    assert abs(l16.item() - l32.item()) < 2e-2 * max(1.0, abs(l32.item()))
    cos = {}
    for n, p in m.named_parameters():
        if p.grad is not None and g32[n].abs().max() > 1e-6 * max(g.abs().max() for g in g32.values()):
            cos[n] = F.cosine_similarity(p.grad.flatten().double(), g32[n].flatten().double(), dim=0).item()
    prior = [v for k, v in cos.items() if k.startswith("conv3d.")]
    print(f"bf16 vs fp32 gradients (randomness branch): prior min cosine {min(prior):.4f}, all min {min(cos.values()):.4f}, mean {np.mean(list(cos.values())):.4f}")
    assert len(prior) == 36 and min(prior) > 0.98 and np.mean(list(cos.values())) > 0.98


def test_dropout_training_mode_is_consistent_between_forward_and_backward():
    """train(): dropout masks are stateless hashes of a per-call seed drawn from torch's RNG.  Same torch seed -> same loss; and
    the directional derivative of that (fixed-mask) loss along a parameter direction agrees with <grad, direction>."""
    cfg = synth.mnist_model_config(frames_length=4, width=64, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, 37, DEV).train()
    m.first_stage_model.eval()
    batch = {k: v.to(DEV) for k, v in synth.synth_batch_mnist(2, 4, seed=37).items()}

    def loss_at():
        torch.manual_seed(99)
        return m(batch)[0]
    l0 = loss_at()
    l0.backward()
    assert abs(loss_at().item() - l0.item()) < 1e-6
    torch.manual_seed(100)
    assert abs(m(batch)[0].item() - l0.item()) > 1e-6                            # another seed: another mask
    m.eval()
    assert abs(m(batch)[0].item() - l0.item()) > 1e-6                            # and eval mode has none
    m.train()
    m.first_stage_model.eval()
    p = m.generate_model.blocks[1].mlp.c_fc.weight
    g = p.grad.clone()
    d = torch.randn(p.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    d = d / d.norm()
    eps = 2e-2
    def shift(a):
        with torch.no_grad():
            p.add_(a * d)
    shift(eps)
    lp = loss_at().item()                    # in grad mode: the training pass (the no-grad pass is the dropout-free evaluation)
    shift(-2 * eps)
    lm = loss_at().item()
    shift(eps)
    fd, an = (lp - lm) / (2 * eps), (g * d).sum().item()
    print(f"directional derivative: finite difference {fd:.6f}, analytic {an:.6f}")
    assert abs(fd - an) < 0.05 * max(abs(an), 1e-3) + 2e-4


def test_bf16_training_folds_equal_the_separate_passes():
    """bf16 train(): the folded passes (residual add + dropout + next LayerNorm in one launch, layernorm_bwd writing the next branch's
    masked operand, padded-taps frame convolutions) against the separate launches they replace (config train_emit / train_taps off),
    same dropout seed: same masks, same loss and gradients up to bf16 rounding of one intermediate."""
    cfg = synth.mnist_model_config(frames_length=4, width=256, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, 41, DEV).train()
    m.first_stage_model.eval()
    m.set_precision("bf16")
    batch = {k: v.to(DEV) for k, v in synth.synth_batch_mnist(2, 4, seed=41).items()}

    def run():
        m.zero_grad(set_to_none=True)
        torch.manual_seed(7)
        loss, _ = m(batch)
        loss.backward()
        return loss.item(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    l1, g1 = run()
    with config.override(train_emit=False, train_taps=False):
        l0, g0 = run()
    assert abs(l1 - l0) < 2e-3 * max(1.0, abs(l0))
    cos = {n: F.cosine_similarity(g1[n].flatten().double(), g0[n].flatten().double(), dim=0).item() for n in g0 if g0[n].abs().max() > 0}
    print(f"folded vs separate passes: loss {l1:.6f} / {l0:.6f}, min gradient cosine {min(cos.values()):.6f}")
    assert set(g1) == set(g0) and min(cos.values()) > 0.999


@pytest.mark.parametrize("T,N,K,ld_dy,ld_x", [(4096, 256, 256, 256, 256), (5000, 512, 256, 512, 256), (16384, 512, 2048, 1536, 2048), (70000, 1536, 512, 1536, 512),
                                               (262144, 1536, 512, 1536, 512), (65536, 512, 2048, 512, 2048), (8192, 256, 512, 256, 512)])
def test_gemm_tn_weight_gradient_against_fp64(T, N, K, ld_dy, ld_x):
    """mage_gemm_tn: dW = dY^T X from row-major bf16 operands through the transposing LDS load (no transposed copies), token slices +
    fixed-order partial sums, and the column sums of dY (bias gradient); ragged token counts, strided operands (a column slice of a wider
    matrix), bitwise repeatable."""
    from mage_amd import ops as o
    g = torch.Generator().manual_seed(T + N)
    dy_full = (torch.randn(T, ld_dy, generator=g) * 0.5).to(DEV).bfloat16()
    x_full = torch.randn(T, ld_x, generator=g).to(DEV).bfloat16()
    c0 = ld_dy - N                                          # the LAST N columns of a wider matrix (the k|v part of dqkv)
    dy = dy_full[:, c0:]
    dW, db = o.gemm_tn(dy, x_full, T=T, N=N, K=K, ld_dy=ld_dy, ld_x=ld_x)
    want = dy.double().t() @ x_full[:, :K].double()
    scale = want.abs().max().item()
    assert (dW.double() - want).abs().max().item() < 2e-5 * scale + 1e-3
    want_b = dy.double().sum(0)
    assert (db.double() - want_b).abs().max().item() < 1e-4 * want_b.abs().max().item() + 1e-3
    dW2, db2 = o.gemm_tn(dy, x_full, T=T, N=N, K=K, ld_dy=ld_dy, ld_x=ld_x)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)
    # whole 64-token slabs run on the one-wave-per-SIMD form (csrc/gemm4.hip, gemm_tn4_kernel): same LDS image, same MFMA order per
    # accumulator, same dot2 order for the bias gradient -> the same bits as the 8-wave kernel (library option gemm_no_4w)
    with config.lib_option("gemm_no_4w", 1):
        dW8, db8 = o.gemm_tn(dy, x_full, T=T, N=N, K=K, ld_dy=ld_dy, ld_x=ld_x)
        dW8n, _ = o.gemm_tn(dy, x_full, T=T, N=N, K=K, ld_dy=ld_dy, ld_x=ld_x, want_bias=False)
    dWn, _ = o.gemm_tn(dy, x_full, T=T, N=N, K=K, ld_dy=ld_dy, ld_x=ld_x, want_bias=False)
    assert torch.equal(dW, dW8) and torch.equal(db, db8) and torch.equal(dWn, dW8n) and torch.equal(dWn, dW)
    cs = o.colsum(dy, T=T, C_=N, ld=ld_dy)
    assert (cs.double() - want_b).abs().max().item() < 1e-4 * want_b.abs().max().item() + 1e-3


def test_sum_partials_fixed_order_vector_and_scalar_kernels():
    """mage_sum_partials (split-K partial products, LayerNorm gamma / beta partials): the 16-byte kernel (four waves over the partials,
    fixed-order LDS sum) and the scalar fallback against an fp64 sum; bitwise repeatable; `accumulate` adds to the output."""
    from mage_amd import ops as o
    g = torch.Generator().manual_seed(12)
    for n_part, n, stride in ((1, 256, 256), (3, 1024, 1024), (4, 4096, 4096), (37, 3 * 512, 2048), (64, 262144, 262144), (7, 250, 252)):
        part = torch.randn(n_part, stride, generator=g).to(DEV)
        want = part[:, :n].double().sum(0)
        out = o.sum_partials(part, torch.empty(n, device=DEV), stride=stride, n_part=n_part, n=n)
        assert (out.double() - want).abs().max().item() < 1e-5 * max(1.0, n_part ** 0.5) * 4
        out2 = o.sum_partials(part, torch.empty(n, device=DEV), stride=stride, n_part=n_part, n=n)
        assert torch.equal(out, out2)
        base = torch.randn(n, generator=g).to(DEV)
        acc = o.sum_partials(part, base.clone(), stride=stride, n_part=n_part, n=n, accumulate=True)
        assert (acc.double() - (want + base.double())).abs().max().item() < 1e-5 * max(1.0, n_part ** 0.5) * 4


def test_attention_probability_dropout_forward_backward_consistent():
    """nn.TransformerEncoderLayer(dropout=p) drops attention PROBABILITIES in train() (mage_model.py:193-199): mage_attention's drop_p /
    drop_seed.  The mask is a stateless hash, so (a) it can be read back by sending v = identity rows, (b) the forward equals
    softmax * mask / (1 - p) @ v, (c) the backward's gradients equal autograd's through that expression with the same mask."""
    from mage_amd import ops as o
    g = torch.Generator().manual_seed(4)
    B, S, Wd, H, p, seed = 3, 32, 64, 2, 0.25, 777
    qkv = torch.randn(B * S, 3 * Wd, generator=g).to(DEV)
    kv_len = torch.tensor([32, 20, 27], dtype=torch.int32, device=DEV)
    geo = dict(n_seq=B, inner=1, nq=S, nk=S, n_head=H, q_outer_stride=S, q_axis_stride=1, kv_outer_stride=S, kv_axis_stride=1, kv_len=kv_len,
               kv_len_div=1)
    ld = dict(ldq=3 * Wd, ldk=3 * Wd, ldv=3 * Wd, ldo=Wd)
    out = o.attention(qkv, qkv[:, Wd:], qkv[:, 2 * Wd:], torch.empty(B * S, Wd, device=DEV), drop_p=p, drop_seed=seed, **ld, **geo)
    out2 = o.attention(qkv, qkv[:, Wd:], qkv[:, 2 * Wd:], torch.empty(B * S, Wd, device=DEV), drop_p=p, drop_seed=seed, **ld, **geo)
    out0 = o.attention(qkv, qkv[:, Wd:], qkv[:, 2 * Wd:], torch.empty(B * S, Wd, device=DEV), **ld, **geo)
    assert torch.equal(out, out2) and not torch.equal(out, out0)
    # read the mask back: v = one-hot rows (key j -> e_j) turns out[i, :] into the dropped probabilities of query i
    eye = torch.zeros(B * S, 3 * Wd, device=DEV)
    eye[:, :2 * Wd] = qkv[:, :2 * Wd]
    for h in range(H):
        eye[:, 2 * Wd + h * 32:2 * Wd + (h + 1) * 32] = torch.eye(32, device=DEV).repeat(B, 1)
    pd = o.attention(eye, eye[:, Wd:], eye[:, 2 * Wd:], torch.empty(B * S, Wd, device=DEV), drop_p=p, drop_seed=seed, **ld, **geo)
    p0 = o.attention(eye, eye[:, Wd:], eye[:, 2 * Wd:], torch.empty(B * S, Wd, device=DEV), **ld, **geo)
    pd, p0 = pd.view(B, S, H, 32).permute(0, 2, 1, 3), p0.view(B, S, H, 32).permute(0, 2, 1, 3)        # [B, H, i, j]
    valid = p0 > 0
    mask = (pd > 0) | ~valid
    keep_rate = mask[valid].float().mean().item()
    assert abs(keep_rate - (1 - p)) < 0.03, keep_rate
    torch.testing.assert_close(pd[mask & valid], p0[mask & valid] / (1 - p), rtol=1e-5, atol=1e-7)
    # autograd through the written-out expression with that mask
    x = qkv.clone().double().requires_grad_(True)
    q, k, v = (x[:, i * Wd:(i + 1) * Wd].view(B, S, H, 32).permute(0, 2, 1, 3) for i in range(3))
    sc = q @ k.transpose(-1, -2) * 32 ** -0.5
    jj = torch.arange(S, device=DEV)
    sc = sc.masked_fill(jj[None, None, None, :] >= kv_len.long()[:, None, None, None], float("-inf"))
    pr = torch.softmax(sc, -1) * mask.double() / (1 - p)
    want = (pr @ v).permute(0, 2, 1, 3).reshape(B * S, Wd)
    torch.testing.assert_close(out.double(), want.detach(), rtol=1e-5, atol=1e-5)
    dout = torch.randn(B * S, Wd, generator=g).to(DEV)
    want.backward(dout.double())
    dqkv = torch.zeros(B * S, 3 * Wd, device=DEV)
    o.attention_bwd(qkv, qkv[:, Wd:], qkv[:, 2 * Wd:], dout, dqkv, dqkv[:, Wd:], dqkv[:, 2 * Wd:], ld_dq=3 * Wd, ld_dk=3 * Wd, ld_dv=3 * Wd,
                    drop_p=p, drop_seed=seed, **ld, **geo)
    torch.testing.assert_close(dqkv.double(), x.grad, rtol=1e-4, atol=1e-5)


def test_flat_adam_training_steps_track_torch_adam_on_the_oracle():
    """Three optimizer steps: HIP forward/backward + FlatAdam (one fused launch over the flat arena) against the oracle's autograd +
    torch.optim.Adam with the reference's hyper-parameters (main_mage.py:121)."""
    from mage_amd.optim import FlatAdam
    L = 4
    cfg = synth.mnist_model_config(frames_length=L, width=64, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, 39, DEV)
    batch = synth.synth_batch_mnist(2, L, seed=39)
    sd = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.startswith("first_stage_model.") else v) for k, v in cpu_sd(m).items()}
    ref_params = [v for v in sd.values() if v.requires_grad]
    ref_opt = torch.optim.Adam(ref_params, lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
    opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
    db = {k: v.to(DEV) for k, v in batch.items()}
    for step in range(3):
        ref_opt.zero_grad()
        ref_loss, _ = O.mage_forward_loss(sd, batch, L)
        ref_loss.backward()
        for v in ref_params:
            if v.grad is None:
                v.grad = torch.zeros_like(v)
        ref_opt.step()
        opt.zero_grad()
        loss, _ = m(db)
        loss.backward()
        opt.step()
        print(f"step {step}: loss {loss.item():.6f} (oracle {ref_loss.item():.6f})")
        assert abs(loss.item() - ref_loss.item()) < 2e-4
    got = m.state_dict()
    worst = max(rel(got[k], v) for k, v in sd.items() if v.requires_grad)
    assert worst < 2e-3, worst                                     # Adam divides by sqrt(v): tiny gradients amplify rounding
    assert all(p.data_ptr() >= opt.flat_p.data_ptr() for p in opt.params)


def test_ddp_wrapped_model_and_rccl_sharded_step():
    """The reference wraps the model in DistributedDataParallel(find_unused_parameters=True) (main_mage.py:95) and steps Adam.
    One rank, backend nccl (= RCCL): the DDP-wrapped HIP model trains (its reducer sees one autograd node feeding every parameter),
    and FlatAdam's reduce-scatter -> shard update -> all-gather path over RCCL gives the same parameters as the unsharded step."""
    import os
    import torch.distributed as dist
    from mage_amd.optim import FlatAdam
    from mage_amd.utils.dist import free_port
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        L = 4
        cfg = synth.mnist_model_config(frames_length=L, width=64, layers=3, vq_dim=32, K=64)
        batch = {k: v.to(DEV) for k, v in synth.synth_batch_mnist(2, L, seed=41).items()}
        results = []
        for mode in ("plain", "ddp+sharded"):
            m = build_mage(cfg, 41, DEV)
            net = m
            if mode != "plain":
                net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
            opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, shard=(mode != "plain"))
            assert opt.sharded == (mode != "plain")
            for _ in range(2):
                opt.zero_grad()
                loss, ld = net(batch)
                loss.backward()
                opt.step()
            results.append((loss.item(), {k: v.clone() for k, v in m.state_dict().items()}))
        assert abs(results[0][0] - results[1][0]) < 1e-6
        for k in results[0][1]:        # not bitwise: the two embedding scatters use fp32 atomics (order varies run to run)
            assert torch.allclose(results[0][1][k].float(), results[1][1][k].float(), atol=2e-6, rtol=1e-5), k
        assert dist.get_backend() == "nccl"
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ stage-1 VQ-VAE training
def test_batchnorm_training_kernels():
    o = ops()
    rows, Cc = 3000, 64
    x = (rnd(rows, Cc, seed=50) * 2 + 1).requires_grad_()
    g, b = (1 + 0.2 * rnd(Cc, seed=51)).requires_grad_(), rnd(Cc, seed=52).requires_grad_()
    res = rnd(rows, Cc, seed=53)
    # training-mode BatchNorm over the rows, spelled out (autograd of plain tensor ops: F.batch_norm's CPU backward on a
    # channel-fastest strided view of these rows was found unreliable while writing this test)
    y = F.relu((x - x.mean(0)) / (x.var(0, unbiased=False) + 1e-5).sqrt() * g + b + res)
    dy = rnd(rows, Cc, seed=54)
    y.backward(dy)
    mean, var, rstd = o.bn_train_stats(x.detach().to(DEV), 1e-5)
    assert rel(mean, x.detach().mean(0)) < 1e-6 and rel(var, x.detach().var(0, unbiased=False)) < 1e-5
    got = o.bn_apply(x.detach().to(DEV), mean, rstd, g.detach().to(DEV), b.detach().to(DEV), torch.empty(rows, Cc, device=DEV), True, residual=res.to(DEV))
    assert rel(got, y) < 2e-6
    dx = torch.empty(rows, Cc, device=DEV)
    dg, db = o.bn_backward(x.detach().to(DEV), dy.to(DEV), mean, rstd, g.detach().to(DEV), dx, mask=got)
    assert rel(dx, x.grad) < 2e-5 and rel(dg, g.grad) < 2e-5 and rel(db, b.grad) < 2e-5


@pytest.mark.parametrize("tag", ["vqvae_f4_train_small", "vqvae_f4_train"])
def test_vqvae_stage1_training_step_matches_the_reference(tag):
    """train_vqvae.py:13-27 on the HIP path: model.train(); x_tilde, z_e_x, z_q_x = model(images); the three-term loss (computed by
    the caller with torch, as the script does); loss.backward().  Loss terms, outputs, EVERY parameter gradient and the BatchNorm
    running buffers against the reference's own step."""
    from tests.helpers import golden, t
    from tests.test_oracle_golden import check_vq_train_grads
    g = golden(tag)
    from tests.helpers import build_vqvae
    m = build_vqvae(1, 4, int(g["dim"]), int(g["K"]), int(g["seed"]), DEV).train()
    x = synth.synth_batch_mnist(int(g["n_img"]), 1, seed=int(g["seed"]))["images"][:, 0].contiguous().to(DEV)
    x_tilde, z_e, z_q = m(x)
    assert x_tilde.requires_grad and z_e.requires_grad and z_q.requires_grad
    rec, vql, com = F.mse_loss(x_tilde, x), F.mse_loss(z_q, z_e.detach()), F.mse_loss(z_e, z_q.detach())
    loss = rec + vql + float(g["beta"]) * com
    assert abs(rec.item() - float(g["rec"])) < 1e-5 and abs(vql.item() - float(g["vq"])) < 1e-5 * max(1, float(g["vq"]))
    assert abs(com.item() - float(g["commit"])) < 1e-5 * max(1, float(g["commit"])) and abs(loss.item() - float(g["loss"])) < 1e-4
    torch.testing.assert_close(x_tilde.detach()[:, :, ::4, ::4].cpu(), t(g["x_tilde_sub"]), atol=1e-4, rtol=0)
    loss.backward()
    grads = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()}
    worst = check_vq_train_grads(g, grads, GRAD_TOL)
    print(f"{tag}: worst relative gradient error {worst:.2e}")
    for n, b in m.named_buffers():
        if n.endswith("running_mean") or n.endswith("running_var"):
            torch.testing.assert_close(b.cpu(), t(g["buf." + n]), atol=2e-5, rtol=1e-4)
        if n.endswith("num_batches_tracked"):
            assert int(b) == 1
    # the eval-mode entry points are untouched by the training pass (and refuse training-mode BatchNorm)
    with pytest.raises(NotImplementedError):
        m.encode(x)
    m.eval()
    assert m.encode(x).dtype == torch.int64


@pytest.mark.parametrize("tag", ["vqvae_f8_train_small", "vqvae_f8_train"])
def test_vqvae8_stage1_training_step_matches_the_reference(tag):
    """train_vqvae.py:13-27 with --dataset cater-gen (down_ratio 8: bottleneck blocks, MaxPool2d, nearest Upsample, 7x7 RGB stem, 1x1
    tanh head, codebook dimension 4*dim) on the HIP path: outputs, loss terms and EVERY parameter gradient against the reference's
    own step."""
    from tests.helpers import build_vqvae, golden, t
    from tests.test_oracle_golden import check_vq_train_grads
    g = golden(tag)
    m = build_vqvae(3, 8, int(g["dim"]), int(g["K"]), int(g["seed"]), DEV).train()
    x = synth.synth_batch_cater(int(g["n_img"]), 1, seed=int(g["seed"]), res=64)["images"][:, 0].contiguous().to(DEV)
    x_tilde, z_e, z_q = m(x)
    assert x_tilde.requires_grad and z_e.requires_grad and z_q.requires_grad and z_e.shape[1] == 4 * int(g["dim"])
    rec, vql, com = F.mse_loss(x_tilde, x), F.mse_loss(z_q, z_e.detach()), F.mse_loss(z_e, z_q.detach())
    loss = rec + vql + float(g["beta"]) * com
    assert abs(rec.item() - float(g["rec"])) < 1e-5 and abs(vql.item() - float(g["vq"])) < 1e-5 * max(1, float(g["vq"]))
    assert abs(com.item() - float(g["commit"])) < 1e-5 * max(1, float(g["commit"])) and abs(loss.item() - float(g["loss"])) < 1e-4 * max(1, float(g["loss"]))
    torch.testing.assert_close(x_tilde.detach()[:, :, ::4, ::4].cpu(), t(g["x_tilde_sub"]), atol=1e-4, rtol=0)
    loss.backward()
    grads = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()}
    worst = check_vq_train_grads(g, grads, GRAD_TOL)
    print(f"{tag}: worst relative gradient error {worst:.2e}")
    m.eval()                                                       # eval mode: values only, the inference path
    with torch.no_grad():
        xt2, _, _ = m(x)
    torch.testing.assert_close(xt2, x_tilde.detach(), atol=1e-5, rtol=0)


def test_vqvae_f4_stage1_training_with_rgb_frames_matches_oracle_autograd():
    """The f4 stack on 3-channel frames (the stem's weight gradient over the image padded to 8 channels, the 48-tap head): every
    gradient against autograd through the oracle's training-mode forward (the reference's scripts only train f4 on 1-channel MNIST)."""
    from tests.helpers import build_vqvae
    m = build_vqvae(3, 4, 32, 64, 23, DEV).train()
    x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(23)) * 2 - 1
    sd = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in cpu_sd(m).items()}
    want_loss, _, _ = O.vqvae_train_loss(sd, "", x, beta=2.0)
    names = [n for n, _ in m.named_parameters()]
    want = dict(zip(names, torch.autograd.grad(want_loss, [sd[n] for n in names], allow_unused=True)))
    xt, ze, zq = m(x.to(DEV))
    loss = F.mse_loss(xt, x.to(DEV)) + F.mse_loss(zq, ze.detach()) + 2.0 * F.mse_loss(ze, zq.detach())
    assert abs(loss.item() - want_loss.item()) < 1e-5
    loss.backward()
    gmax = max(g.abs().max().item() for g in want.values() if g is not None)
    worst = 0.0
    for n, p in m.named_parameters():
        g = want[n] if want[n] is not None else torch.zeros_like(sd[n])
        if g.abs().max().item() < 1e-4 * gmax:                     # biases in front of a training-mode BatchNorm: zero gradient, noise
            assert p.grad.abs().max().item() < 1e-4 * gmax, n
            continue
        worst = max(worst, rel(p.grad, g))
    print(f"f4 RGB stage-1 step: worst relative gradient error {worst:.2e}")
    assert worst < GRAD_TOL


def test_pooling_backward_kernels():
    o = ops()
    N, H, W, Cc = 2, 8, 12, 8
    x = rnd(N, Cc, H, W, seed=1)
    x[0, :, 0, 0] = x[0, :, 0, 1]                                  # a tie inside a window: the first maximum takes the gradient
    x = x.requires_grad_()
    dy = rnd(N, Cc, H // 2, W // 2, seed=2)
    F.max_pool2d(x, 2).backward(dy)
    rows = lambda t_: t_.permute(0, 2, 3, 1).reshape(-1, Cc).contiguous().to(DEV)
    dx = o.maxpool2_bwd(rows(x.detach()), rows(dy), N=N, H=H, W=W, Cc=Cc)
    assert torch.equal(dx.view(N, H, W, Cc).permute(0, 3, 1, 2).cpu(), x.grad)
    u = rnd(N, Cc, H, W, seed=3).requires_grad_()
    du = rnd(N, Cc, 2 * H, 2 * W, seed=4)
    F.interpolate(u, scale_factor=2, mode="nearest").backward(du)
    dxu = o.upsample2_bwd(rows(du), N=N, H=H, W=W, Cc=Cc)
    torch.testing.assert_close(dxu.view(N, H, W, Cc).permute(0, 3, 1, 2).cpu(), u.grad, atol=1e-6, rtol=1e-6)


def test_vqvae_stage1_training_loop_reduces_the_loss():
    """A few steps of train_vqvae.py's loop (Adam lr 1e-4 as its default is too slow to show in 5 steps: 1e-3) with FlatAdam."""
    from mage_amd.optim import FlatAdam
    from tests.helpers import build_vqvae
    m = build_vqvae(1, 4, 32, 64, 15, DEV).train()
    opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    x = synth.synth_batch_mnist(8, 1, seed=15)["images"][:, 0].contiguous().to(DEV)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        x_tilde, z_e, z_q = m(x)
        loss = F.mse_loss(x_tilde, x) + F.mse_loss(z_q, z_e.detach()) + 2.0 * F.mse_loss(z_e, z_q.detach())
        loss.backward()
        opt.step()
        losses.append(loss.item())
    print("stage-1 losses:", [round(l, 4) for l in losses])
    assert losses[-1] < losses[0] and all(np.isfinite(losses))


def oracle_grads_random(sd, batch, L, eps, alpha, beta):
    sd = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.startswith("first_stage_model.") else v) for k, v in sd.items()}
    loss, parts, _, _ = O.mage_forward_loss_random(sd, batch, L, eps, alpha=alpha, beta=beta)
    names = [k for k, v in sd.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    return loss.item(), parts, {k: g for k, g in zip(names, gs)}


def prior_relu_margin(sd, tok=None, lat=None):
    """Smallest |pre-activation| over the ReLUs of the Conv3d video prior (BasicBlock, mage_model.py:280-297), float64; the prior's
    input is the token embedding of tok [B, L, h, w] or (MAGE+) the Linear embedding of lat [B, L, E, h, w]."""
    sd = {k: v.double() for k, v in sd.items() if k.startswith(("conv3d.", "visual_token_embedding."))}
    if tok is not None:
        v = sd["visual_token_embedding.weight"][tok].permute(0, 4, 1, 2, 3).contiguous()
    else:
        v = F.linear(lat.double().permute(0, 1, 3, 4, 2), sd["visual_token_embedding.weight"], sd["visual_token_embedding.bias"])
        v = v.permute(0, 4, 1, 2, 3).contiguous()
    mn = float("inf")
    for i in range(4):
        p = f"conv3d.{i}."
        t1 = F.group_norm(F.conv3d(v, sd[p + "conv1.weight"], None, stride=(2, 1, 1), padding=1), 16, sd[p + "bn1.weight"], sd[p + "bn1.bias"])
        o = F.group_norm(F.conv3d(F.relu(t1), sd[p + "conv2.weight"], None, padding=1), 16, sd[p + "bn2.weight"], sd[p + "bn2.bias"])
        r = F.group_norm(F.conv3d(v, sd[p + "downsample.0.weight"], None, stride=(2, 1, 1), padding=1), 16, sd[p + "downsample.1.weight"],
                         sd[p + "downsample.1.bias"])
        mn = min(mn, t1.abs().min().item(), (o + r).abs().min().item())
        v = F.relu(o + r)
    return mn


@pytest.mark.parametrize("B,L,seed,beta,alpha", [(2, 9, 42, 0.00025, 0.001), (1, 12, 52, 0.5, 0.1)])
def test_loss_backward_with_the_randomness_branch_matches_oracle_autograd(B, L, seed, beta, alpha):
    """config/mage_caterv1.yaml's family (use_cids=True, randomness=True: Conv3d video prior -> reparameterisation + KL -> ADAIN, speed
    l2 term) at small width: loss.backward() on the HIP path (fp32, eval) against autograd through the oracle, every trainable
    parameter within 1e-4 of its tensor's largest reference entry.  The second case weights the KL and l2 terms up so that their
    gradients are not hidden under the reconstruction term's.

    The prior has ~2e5 ReLU inputs of unit scale here, so some pre-activation always lies within a few 1e-6 of zero -- the size of the
    fp32 forward difference between two correct implementations.  Where the two disagree on such a sign, ONE flipped mask element moves
    the weight gradients of every earlier layer by 1e-3..1e-2 (measured, tools/prior_bwd_debug.py: random tokens 1e-6, a batch with
    |t| = 1.4e-6: 5e-3) -- the gradient analogue of an arg-min tie.  The seeds are the ones whose smallest |t| is >= 1e-5 (asserted)."""
    cfg = synth.cater_model_config(frames_length=L, width=64, layers=3, vq_dim=32, K=64)
    cfg["params"]["beta"], cfg["params"]["alpha"] = beta, alpha
    m = build_mage(cfg, seed, DEV)
    batch = synth.synth_batch_cater(B, L, seed=seed, text_len=9)
    eps = torch.randn(B, 64, 16, 16, generator=torch.Generator().manual_seed(seed))
    sd = cpu_sd(m)
    tok = O.vqvae_encode(sd, "first_stage_model.", batch["images"].reshape(B * L, *batch["images"].shape[2:])).view(B, L, 16, 16)
    assert prior_relu_margin(sd, tok) >= 1e-5
    want_loss, want_parts, want = oracle_grads_random(sd, batch, L, eps, alpha, beta)
    db = {k: v.to(DEV) for k, v in batch.items()}
    db["reparam_noise"] = eps.to(DEV)
    loss, ld = m(db)
    assert loss.requires_grad and abs(loss.item() - want_loss) < 1e-4 * max(1.0, abs(want_loss))
    assert abs(ld["val/kl_loss"] - want_parts["kl_loss"]) < 1e-4 * max(1.0, abs(want_parts["kl_loss"]))
    assert abs(ld["val/prediction"] - want_parts["prediction"]) < 1e-4
    loss.backward()
    gmax = max(g.abs().max().item() for g in want.values() if g is not None)
    worst, n_checked, prior_checked = ("", 0.0), 0, 0
    for name, p in m.named_parameters():
        if name.startswith("first_stage_model."):
            assert p.grad is None
            continue
        g_ref = want.get(name)
        assert p.grad is not None, name
        if g_ref is None or g_ref.abs().max().item() == 0.0:
            assert p.grad.abs().max().item() == 0.0, name
            continue
        if name == "ma_encoder.blocks.0.mlp.c_proj.bias":
            # a per-channel constant in front of ADAIN's instance norm (the MA encoder's last bias): the exact gradient is zero, the
            # reference's is rounding noise; bounded against the scale of the real gradients instead
            assert g_ref.abs().max().item() < 1e-6 * gmax and p.grad.abs().max().item() < 1e-6 * gmax, name
            continue
        r = rel(p.grad, g_ref)
        n_checked += 1
        prior_checked += name.startswith(("conv3d.", "adain.", "conv_mu2", "conv_var2", "conv_d2"))
        if r > worst[1]:
            worst = (name, r)
    print(f"{n_checked} gradients checked ({prior_checked} of the randomness branch), worst relative error {worst[1]:.2e} at {worst[0]}")
    assert worst[1] < GRAD_TOL, worst
    assert prior_checked == 4 * 9 + 8 + 4 + 1 and n_checked >= 130


@pytest.mark.parametrize("B,L,seed,min_margin", [(1, 10, 69, 1e-5), (2, 9, 62, 5e-6)])
def test_loss_backward_mage_plus_matches_oracle_autograd(B, L, seed, min_margin):
    """config/mage+_*.yaml's family (use_cids=False: Linear embedding of the first stage's latents, the ln_q / ln_kv TransformerBlock
    variant, GroupNorm(32) + SiLU + Conv3d head, MSE loss, randomness branch with the PID-controlled beta) over the stand-in latent first
    stage: loss.backward() on the HIP path (fp32, eval) against autograd through the oracle.  Seeds chosen for their ReLU margin, see
    test_loss_backward_with_the_randomness_branch_matches_oracle_autograd."""
    from mage_amd.modules.mage_model import PIDControl
    cfg = synth.magep_model_config(frames_length=L, width=64, layers=3)
    m = build_mage(cfg, seed, DEV)
    assert m.ma_encoder.mage_plus
    batch = synth.synth_batch_cater(B, L, seed=seed, text_len=9, vocab=50)
    eps = torch.randn(B, 64, 16, 16, generator=torch.Generator().manual_seed(seed))
    lat = m.first_stage_model.encode(batch["images"].reshape(B * L, *batch["images"].shape[2:])).view(B, L, 4, 16, 16).cpu()
    sd0 = cpu_sd(m)
    assert prior_relu_margin(sd0, lat=lat) >= min_margin
    sd = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.startswith("first_stage_model.") else v) for k, v in sd0.items()}
    final, parts, pred = O.mage_forward_loss_latent(sd, batch, L, lat, eps, v_kl=cfg["params"]["v_kl"], pid=PIDControl(), mage_plus=True)
    names = [k for k, v in sd.items() if v.requires_grad]
    want = dict(zip(names, torch.autograd.grad(final, [sd[k] for k in names], allow_unused=True)))
    db = {k: v.to(DEV) for k, v in batch.items()}
    db["reparam_noise"] = eps.to(DEV)
    loss, ld = m(db)
    assert loss.requires_grad and abs(loss.item() - final.item()) < 1e-4 * max(1.0, abs(final.item()))
    assert abs(ld["val/prediction"] - parts["prediction"]) < 1e-4 * max(1.0, parts["prediction"]) and abs(ld["val/beta"] - parts["beta"]) < 1e-7
    torch.testing.assert_close(m.last_logits.view(B, L - 1, 16, 16, -1)[..., :4].cpu(), pred.detach(), atol=1e-4, rtol=1e-4)
    loss.backward()
    gmax = max(g.abs().max().item() for g in want.values() if g is not None)
    worst, n_checked, seen = ("", 0.0), 0, set()
    for name, p in m.named_parameters():
        g_ref = want.get(name)
        if name.startswith("first_stage_model.") or g_ref is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, name
            continue
        assert p.grad is not None, name
        if name == "ma_encoder.blocks.0.mlp.c_proj.bias":                   # a constant in front of ADAIN's instance norm: zero gradient
            assert g_ref.abs().max().item() < 1e-6 * gmax and p.grad.abs().max().item() < 1e-6 * gmax, name
            continue
        r = rel(p.grad, g_ref)
        n_checked += 1
        seen.add(name)
        if r > worst[1]:
            worst = (name, r)
    print(f"{n_checked} gradients checked, worst relative error {worst[1]:.2e} at {worst[0]}")
    assert worst[1] < GRAD_TOL, worst
    for must in ("visual_token_embedding.weight", "visual_token_embedding.bias", "generate_model.out.0.weight", "generate_model.out.2.weight",
                 "generate_model.out.2.bias", "ma_encoder.blocks.0.ln_q.weight", "ma_encoder.blocks.0.ln_kv.bias", "conv3d.0.conv1.weight",
                 "adain.conv_mu.0.weight", "conv_d2.weight"):
        assert must in seen, must


@pytest.mark.parametrize("B,Cc,P", [(1, 64, 256), (2, 64, 256), (3, 512, 64)])
def test_adain_backward(B, Cc, P):
    """mage_adain_bwd against autograd of gamma * instance_norm(x) + beta (ADAIN2D, mage_model.py:299-314)."""
    o = ops()
    x = (rnd(B, P, Cc, seed=1) * 2 + 0.5).requires_grad_()
    gam, bet = rnd(B, P, Cc, seed=2).requires_grad_(), rnd(B, P, Cc, seed=3).requires_grad_()
    dout = rnd(B, P, Cc, seed=4)
    # the instance norm written out: F.instance_norm's CPU backward is wrong for a batch of one
    xn = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    (gam * xn + bet).backward(dout)
    dx, dg = o.adain_bwd(x.detach().reshape(B * P, Cc).to(DEV), gam.detach().reshape(B * P, Cc).to(DEV), dout.reshape(B * P, Cc).to(DEV),
                         B=B, P=P, Cc=Cc)
    assert rel(dx.view(B, P, Cc), x.grad) < 2e-5 and rel(dg.view(B, P, Cc), gam.grad) < 2e-5


@pytest.mark.parametrize("B,Cc,groups,rows,act,with_res", [(1, 64, 16, 300, 1, True), (2, 64, 16, 512, 1, False), (3, 512, 32, 96, 2, False),
                                                            (2, 128, 16, 64, 0, True)])
def test_groupnorm_backward_with_row_maps(B, Cc, groups, rows, act, with_res):
    """mage_groupnorm_bwd against autograd of act(group_norm(x) + residual), with x / dy living in padded row maps (the frame buffers of
    the Conv3d video prior): rows outside the map are not touched."""
    o = ops()
    xs, xo, ds, do_ = rows + 40, 8, rows + 24, 16
    x = (rnd(B, rows, Cc, seed=1) * 1.5 + 0.3).requires_grad_()
    res = rnd(B, rows, Cc, seed=2).requires_grad_() if with_res else None
    g, b = (1 + 0.2 * rnd(Cc, seed=3)).requires_grad_(), (0.1 * rnd(Cc, seed=4)).requires_grad_()
    dy = rnd(B, rows, Cc, seed=5)
    t_ = F.group_norm(x.permute(0, 2, 1), groups, g, b, 1e-5).permute(0, 2, 1)
    if with_res:
        t_ = t_ + res
    y = F.relu(t_) if act == 1 else F.silu(t_) if act == 2 else t_
    y.backward(dy)
    xbuf = torch.full((B * xs + 64, Cc), 7.0)
    dybuf = torch.full((B * ds + 64, Cc), 9.0)
    for i in range(B):
        xbuf[i * xs + xo:i * xs + xo + rows] = x.detach()[i]
        dybuf[i * ds + do_:i * ds + do_ + rows] = dy[i]
    xd, dyd = xbuf.to(DEV), dybuf.to(DEV)
    geo = dict(n_samples=B, rows_per_sample=rows, sample_stride_rows=xs, row_off=xo, groups=groups)
    stats = torch.empty(B, groups, 2, device=DEV)
    resd = res.detach().reshape(B * rows, Cc).to(DEV) if with_res else None
    yd = o.groupnorm_act(xd, g.detach().to(DEV), b.detach().to(DEV), torch.empty(B * rows, Cc, device=DEV), eps=1e-5, act=act, residual=resd,
                         stats=stats, **geo)
    assert rel(yd.view(B, rows, Cc), y) < 1e-5
    dx = torch.full_like(xd, -3.0)
    dg, db, dres = o.groupnorm_bwd(xd, g.detach().to(DEV), b.detach().to(DEV), stats, dyd, dx, act=act, residual=resd, dy_sample_stride_rows=ds,
                                   dy_row_off=do_, want_dres=with_res, **geo)
    dxc = dx.cpu()
    for i in range(B):
        assert rel(dxc[i * xs + xo:i * xs + xo + rows], x.grad[i]) < 2e-5
        dxc[i * xs + xo:i * xs + xo + rows] = -3.0
    assert torch.equal(dxc, torch.full_like(dxc, -3.0))                      # nothing outside the row map was written
    assert rel(dg, g.grad) < 2e-5 and rel(db, b.grad) < 2e-5
    if with_res:
        assert rel(dres.view(B, rows, Cc), res.grad) < 2e-5


def test_reparam_kl_backward():
    o = ops()
    B, n = 3, 1000
    mu, lv = rnd(B, n, seed=1).requires_grad_(), (0.5 * rnd(B, n, seed=2)).requires_grad_()
    eps, dz = rnd(B, n, seed=3), rnd(B, n, seed=4)
    z = eps * (0.5 * lv).exp() + mu
    kl = -0.5 * torch.mean(torch.sum(1 + lv - mu.pow(2) - lv.exp(), dim=1))
    ((z * dz).sum() + 0.37 * kl).backward()
    coef = torch.tensor([0.37 / B], device=DEV)
    dmu, dlv = o.reparam_kl_bwd(mu.detach().to(DEV), lv.detach().to(DEV), eps.to(DEV), dz.to(DEV), coef)
    assert rel(dmu, mu.grad) < 1e-5 and rel(dlv, lv.grad) < 1e-5
