"""GPU tests of the FAST PARITY modes ('f16x3' / 'bf16x3'): split-precision operands (x ~ hi + lo in two 16-bit pieces, three
MFMA products per K slab, include/mage_hip.h MAGE_F16X3 / MAGE_BF16X3).

Unit level: the split representation, the split GEMM (lockstep and 8-phase kernels, residual / QuickGELU / split-output
epilogues) against an fp64 product, and every producer of split rows (LayerNorm, attention, embedding) against its fp32 twin.
Model level: the SAME golden gates as the fp32 mode (tests/test_gpu_parity.py) -- reference token sequences bit-exact
(n_soft == 0), logits / frames within 1e-4 (north_star) -- for 'f16x3', and incremental == full loop bitwise for both kinds."""
import numpy as np
import pytest
import torch

from mage_amd import ops
from mage_amd.utils import synth
from tests.helpers import assert_tokens, build_mage, chk, golden, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOK_TOL = 2e-5
LOGIT_TOL = 1e-4
KINDS = [pytest.param(ops.F16X3, id="f16x3"), pytest.param(ops.BF16X3, id="bf16x3")]
# measured worst |error| of a K = 512..2048 product of unit-scale operands (tools/split_probe.py, profiles/r03_split_probe.txt):
# f16x3 5e-6 .. 9e-6 (the exact-fp32 MFMA chain: 9e-6 .. 2e-5), bf16x3 2.3e-5 .. 2.6e-5
GEMM_TOL = {ops.F16X3: 2e-5, ops.BF16X3: 6e-5}


def unsplit(y, kind):
    rows, c2 = y.shape
    v = y.view(rows, c2 // 128, 2, 64).float()
    lo = v[:, :, 1] / (2048.0 if kind == ops.F16X3 else 1.0)
    return (v[:, :, 0].double() + lo.double()).reshape(rows, c2 // 2)


def dev_batch(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


@pytest.mark.parametrize("kind", KINDS)
def test_split_representation(kind):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(300, 256, generator=g) * torch.logspace(-3, 3, 256)[None, :]).to(DEV)
    s = ops.split(x, kind)
    assert tuple(s.shape) == (300, 512) and s.dtype == ops.split_dtype(kind)
    # f16 pieces: 2^-22 relative down to the f16 normal range (6e-5), an absolute floor of 2^-36 below it (subnormal pieces)
    err = (unsplit(s, kind) - x.double()).abs()
    bound = x.double().abs() * (2.0 ** -21 if kind == ops.F16X3 else 2.0 ** -17) + (2.0 ** -35 if kind == ops.F16X3 else 0.0)
    assert bool((err <= bound).all()), (err / x.double().abs()).max().item()
    z = ops.split(torch.zeros(4, 64, device=DEV), kind)
    assert not z.view(torch.int16).any()


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("shape", [(2048, 512, 512), (768, 1536, 512), (1000, 192, 64), (131072, 512, 2048), (131072, 2048, 512)])
def test_split_gemm_against_fp64(kind, shape):
    """y = r + a w^T + b and the split-output QuickGELU form; small shapes run the lockstep kernel (incl. ragged M), the large
    ones the 8-phase kernel."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(DEV)
    rows = slice(0, min(M, 2048))
    a_s, w_s = ops.split(a, kind), ops.split(w, kind)
    y = ops.gemm(a_s, w_s, torch.empty(M, N, device=DEV), M=M, N=N, K=K, lda=2 * K, ldy=N, bias=b, residual=r, ldr=N, split_kind=kind)
    want = a[rows].double() @ w.double().t() + b.double() + r[rows].double()
    assert (y[rows].double() - want).abs().max().item() < GEMM_TOL[kind]
    tail = slice(M - 300, M)
    want_t = a[tail].double() @ w.double().t() + b.double() + r[tail].double()
    assert (y[tail].double() - want_t).abs().max().item() < GEMM_TOL[kind]
    ys = ops.gemm(a_s, w_s, ops.split_empty(M, N, kind, DEV), M=M, N=N, K=K, lda=2 * K, ldy=2 * N, bias=b, act=ops.ACT_QUICKGELU,
                  split_kind=kind, y_split=True)
    pre = a[rows].double() @ w.double().t() + b.double()
    assert (unsplit(ys[rows], kind) - pre * torch.sigmoid(1.702 * pre)).abs().max().item() < GEMM_TOL[kind]


@pytest.mark.parametrize("kind", KINDS)
def test_split_gemm_row_slices_match_whole(kind):
    """Row-block invariance across kernels: the first 2048 rows computed alone (lockstep kernel, 128-row tiles) are bitwise the
    first 2048 rows of the 131072-row launch (8-phase kernel): what keeps incremental decoding == full loop."""
    M, N, K = 131072, 512, 512
    g = torch.Generator().manual_seed(7)
    a = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    a_s, w_s = ops.split(a, kind), ops.split(w, kind)
    y = ops.gemm(a_s, w_s, torch.empty(M, N, device=DEV), M=M, N=N, K=K, lda=2 * K, ldy=N, bias=b, split_kind=kind)
    y2 = ops.gemm(a_s[:2048], w_s, torch.empty(2048, N, device=DEV), M=2048, N=N, K=K, lda=2 * K, ldy=N, bias=b, split_kind=kind)
    assert torch.equal(y[:2048], y2)


@pytest.mark.parametrize("kind", KINDS)
def test_split_row_producers_match_fp32_twins(kind):
    tol = 2.0 ** -19 if kind == ops.F16X3 else 2.0 ** -15
    g = torch.Generator().manual_seed(3)
    # LayerNorm
    x = (torch.randn(1024, 512, generator=g) * 3 + 0.5).to(DEV)
    gm, bt = torch.randn(512, generator=g).to(DEV), torch.randn(512, generator=g).to(DEV)
    y32 = ops.layernorm(x, gm, bt, torch.empty_like(x), 1e-5)
    ys = ops.layernorm(x, gm, bt, ops.split_empty(1024, 512, kind, DEV), 1e-5, split_kind=kind)
    assert (unsplit(ys, kind) - y32.double()).abs().max().item() < tol * y32.abs().max().item()
    # attention (H axis geometry of the decoder: 16 rows x 16 heads)
    Cc, H, n_seq, n = 512, 16, 64, 16
    qkv = torch.randn(n_seq * n, 3 * Cc, generator=g).to(DEV)
    geo = dict(ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc, n_seq=n_seq, inner=1, nq=n, nk=n, n_head=H, q_outer_stride=n, q_axis_stride=1,
               kv_outer_stride=n, kv_axis_stride=1)
    o32 = ops.attention(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], torch.empty(n_seq * n, Cc, device=DEV), ldo=Cc, **geo)
    os_ = ops.attention(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], ops.split_empty(n_seq * n, Cc, kind, DEV), ldo=2 * Cc, out_split=kind, **geo)
    assert (unsplit(os_, kind) - o32.double()).abs().max().item() < tol * o32.abs().max().item()
    # embedding into a zero-padded frame buffer
    R, P, nimg = 16, 18, 3
    ids = torch.randint(0, 512, (nimg * R * R,), generator=g).to(DEV)
    table = torch.randn(512, Cc, generator=g).to(DEV)
    p32 = torch.zeros(nimg * P * P + 1, Cc, device=DEV)
    ops.embedding(ids, table, p32, group=R * R, group_stride=P * P, off=P + 1, inner=R, inner_stride=P)
    ps = ops.split_empty(nimg * P * P + 1, Cc, kind, DEV, zero=True)
    ops.embedding(ids, table, ps, group=R * R, group_stride=P * P, off=P + 1, inner=R, inner_stride=P, split_kind=kind)
    assert (unsplit(ps, kind) - p32.double()).abs().max().item() < tol * table.abs().max().item()


@pytest.mark.parametrize("nq,nk,causal", [(16, 16, False), (16, 16, True), (24, 24, True), (32, 32, False), (2, 20, True), (1, 9, True)])
def test_split_attention_on_the_matrix_cores_against_fp32_attention(nq, nk, causal):
    """attention_mfma_split_kernel (q, k, v, out as f16x3 split rows; one and two key blocks; the incremental step's query-block-appended-
    to-a-cache shape) against the fp32 thread-per-query kernel on the same values."""
    kind = ops.F16X3
    g = torch.Generator().manual_seed(nq * 100 + nk)
    Cc, H, n_seq = 512, 16, 48
    q = torch.randn(n_seq * nq, Cc, generator=g).to(DEV)
    kv = torch.randn(n_seq * nk, 2 * Cc, generator=g).to(DEV)
    geo = dict(n_seq=n_seq, inner=1, nq=nq, nk=nk, n_head=H, q_outer_stride=nq, q_axis_stride=1, kv_outer_stride=nk, kv_axis_stride=1, causal=causal)
    want = ops.attention(q, kv, kv[:, Cc:], torch.empty(n_seq * nq, Cc, device=DEV), ldq=Cc, ldk=2 * Cc, ldv=2 * Cc, ldo=Cc, **geo)
    qs, kvs = ops.split(q, kind), ops.split(kv, kind)
    got = ops.attention(qs, kvs, kvs[:, 2 * Cc:], ops.split_empty(n_seq * nq, Cc, kind, DEV), ldq=2 * Cc, ldk=4 * Cc, ldv=4 * Cc, ldo=2 * Cc,
                        out_split=kind, split_kind=kind, **geo)
    err = (unsplit(got, kind) - want.double()).abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("kind", KINDS)
def test_frame_features_and_decoder_pass_track_fp32_mode(kind):
    """Teacher-forced on the reference's own L = 16 tokens: frame convolution + decoder stack in the split mode against the
    exact-fp32 mode of the same kernels and against the reference's logits."""
    g = golden("mage_mnist_L16")
    m = build_mage(synth.mnist_model_config(frames_length=16), int(g["seed"]), DEV)
    B = int(g["B"])
    batch = dev_batch(synth.synth_batch_mnist(B, 16, seed=int(g["seed"])))
    tok0 = m.first_stage_encode(batch["images"][:, 0:1])[:, 0].reshape(B, 1, 256)
    cur = torch.cat([tok0, t(g["gen_tokens"]).long().to(DEV).view(B, -1, 256)[:, :-1]], 1).contiguous()
    ma = m._motion_anchor(tok0.reshape(B, -1), batch, None)
    f32 = m._frame_features(cur, torch.float32)
    lg32 = m.generate_model._run(ma, f32, B=B, hh=16, ww=16)
    m.set_precision("f16x3" if kind == ops.F16X3 else "bf16x3")
    fs = m._frame_features(cur, torch.float32, split=True)
    assert fs.dtype == ops.split_dtype(kind)
    assert (unsplit(fs, kind) - f32.double()).abs().max().item() < GEMM_TOL[kind]
    lgs = m.generate_model._run(ma, fs, B=B, hh=16, ww=16)
    err = (lgs - lg32).abs().max().item()
    ref = t(g["step_logits_sub"])
    err_ref = (lgs.view(B, 15, 16, 16, -1)[:, :, ::8, ::8, ::4].cpu() - ref).abs().max().item()
    print(f"split kind {kind}: teacher-forced logits vs fp32 mode {err:.2e}, vs the reference {err_ref:.2e}")
    assert err_ref < LOGIT_TOL


@pytest.mark.parametrize("tag", ["mage_mnist_L4", "mage_mnist_L6_ragged"])
def test_f16x3_passes_the_fp32_golden_gates(tag):
    g = golden(tag)
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    m = build_mage(synth.mnist_model_config(frames_length=L), seed, DEV).set_precision("f16x3")
    db = dev_batch(synth.synth_batch_mnist(B, L, seed=seed, digits=int(g["digits"]), text_len=int(g["text_len"]), ragged_text=bool(g["ragged"])))
    video = m.autoregressive_generate(db)
    n_soft = assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "AR tokens f16x3")
    assert n_soft == 0, f"{n_soft} argmax flips inside fp32 rounding noise"
    torch.testing.assert_close(video.cpu(), t(g["video"]), atol=LOGIT_TOL, rtol=0)
    torch.testing.assert_close(m.last_logits[:, :, ::4, ::4].cpu(), t(g["step_logits_sub"]), atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(chk(m.last_logits.cpu()), g["step_logits_chk"], rtol=1e-4)
    loss, _ = m(db)
    assert abs(loss.item() - float(g["loss"])) < 1e-4


def test_f16x3_reference_token_sequence_L16():
    """The reference's 16-frame token sequence (7680 free-running argmax decisions, minimum top-2 margin 3.3e-6) in f16x3 mode."""
    g = golden("mage_mnist_L16")
    m = build_mage(synth.mnist_model_config(frames_length=16), int(g["seed"]), DEV).set_precision("f16x3")
    v = m.autoregressive_generate(dev_batch(synth.synth_batch_mnist(int(g["B"]), 16, seed=int(g["seed"]))))
    assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "AR tokens L16 f16x3") == 0
    torch.testing.assert_close(m.last_logits[:, :, ::8, ::8, ::4].cpu(), t(g["step_logits_sub"]), atol=LOGIT_TOL, rtol=0)
    torch.testing.assert_close(v[:, :, :, ::2, ::2].cpu(), t(g["video_sub"]), atol=LOGIT_TOL, rtol=0)


def test_f16x3_reduced_width_and_cater_goldens():
    g = golden("mage_small_d64")
    cfg = synth.mnist_model_config(frames_length=int(g["L"]), width=64, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, int(g["seed"]), DEV).set_precision("f16x3")
    batch = synth.synth_batch_mnist(int(g["B"]), int(g["L"]), seed=int(g["seed"]), text_len=int(g["text_len"]), ragged_text=True)
    video = m.autoregressive_generate(dev_batch(batch))
    assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "AR tokens d64 f16x3") == 0
    torch.testing.assert_close(m.last_logits.cpu(), t(g["step_logits"]), atol=LOGIT_TOL, rtol=0)
    torch.testing.assert_close(video.cpu(), t(g["video"]), atol=LOGIT_TOL, rtol=0)
    g = golden("mage_cater_fullwidth")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    m = build_mage(synth.cater_model_config(frames_length=L), seed, DEV).set_precision("f16x3")
    db = dev_batch(synth.synth_batch_cater(B, L, seed=seed, text_len=int(g["text_len"])))
    db["video_noise"] = t(g["noise"]).to(DEV)
    v = m.autoregressive_generate(db)
    assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "AR tokens cater f16x3") == 0
    torch.testing.assert_close(m.last_logits[:, :, ::4, ::4].cpu(), t(g["step_logits_sub"]), atol=LOGIT_TOL, rtol=0)
    torch.testing.assert_close(v[..., ::4, ::4].cpu(), t(g["video_sub"]), atol=LOGIT_TOL, rtol=0)


def test_f16x3_mage_plus_latent_path_golden():
    g = golden("mage_plus_small")
    B, L = int(g["B"]), int(g["L"])
    m = build_mage(synth.magep_model_config(frames_length=L, width=64, layers=3), int(g["seed"]), DEV).set_precision("f16x3")
    m.ma_encoder.mage_plus = False
    batch = dev_batch(synth.synth_batch_cater(B, L, seed=int(g["seed"]), text_len=int(g["text_len"]), vocab=50))
    batch["video_noise"] = t(g["noise"]).to(DEV)
    video = m.autoregressive_generate(batch)
    torch.testing.assert_close(m.last_logits.cpu(), t(g["pred_latents"]), atol=LOGIT_TOL, rtol=0)
    torch.testing.assert_close(video[..., ::4, ::4].cpu(), t(g["video_sub"]), atol=LOGIT_TOL, rtol=0)


@pytest.mark.parametrize("precision", ["f16x3", "bf16x3"])
def test_split_modes_incremental_is_bit_identical_to_full_loop(precision):
    g = golden("mage_mnist_L6_ragged")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    m = build_mage(synth.mnist_model_config(frames_length=L), seed, DEV).set_precision(precision)
    batch = dev_batch(synth.synth_batch_mnist(B, L, seed=seed, digits=int(g["digits"]), text_len=int(g["text_len"]), ragged_text=True))
    v_full = m.autoregressive_generate(batch)
    t_full = m.last_tokens.clone()
    m.ar_mode = "incremental"
    v_inc = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, t_full) and torch.equal(v_inc, v_full)
    if precision == "f16x3":
        assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "incremental f16x3 tokens") == 0


def test_f16x3_full_size_cfg2_properties():
    """BASELINE cfg2 (B = 64, L = 16) in the fast parity mode: the 8-phase split kernels at full size -- determinism, shard ==
    slice, incremental == full loop (bitwise), and agreement with the exact-fp32 mode's token sequence."""
    m = build_mage(synth.mnist_model_config(frames_length=16), 0, DEV).set_precision("f16x3")
    batch = dev_batch(synth.synth_batch_mnist(64, 16, seed=3))
    v1 = m.autoregressive_generate(batch)
    tok1 = m.last_tokens.clone()
    half = {k: v[32:] for k, v in batch.items()}
    vh = m.autoregressive_generate(half)
    assert torch.equal(m.last_tokens, tok1[32:]) and torch.equal(vh, v1[32:])
    m.ar_mode = "incremental"
    v_inc = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, tok1) and torch.equal(v_inc, v1)
    m.set_precision("fp32")                      # incremental fp32 == full fp32 (tests/test_gpu_parity.py): the cheap way to the fp32 tokens
    m.autoregressive_generate(batch)
    agree = (m.last_tokens == tok1).float().mean().item()
    clips = (m.last_tokens == tok1).flatten(1).all(1).float().mean().item()
    print(f"f16x3 vs fp32 mode, free-running at cfg2: token agreement {agree:.4f}, identical clips {clips:.3f}")
    assert agree > 0.9


def test_f16x3_decoder_against_the_exact_fp32_decoder_and_the_golden():
    """The f4 decoder on f16x3 operands (VectorQuantizedVAE.set_precision('f16x3'): table sum for the first 3x3 convolution, split GEMMs for
    the other 256-channel convolutions) against the exact-fp32 gather path on 64 frames and against the reference's golden frames."""
    from tests.helpers import build_vqvae
    g = golden("vqvae_f4")
    m = build_vqvae(1, 4, 256, 512, int(g["seed"]), DEV)
    ids = torch.randint(0, 512, (64, 16, 16), generator=torch.Generator().manual_seed(2)).to(DEV)
    want = m.decode(ids)
    m.set_precision("f16x3")
    got = m.decode(ids)
    err = (got - want).abs().max().item()
    print(f"f16x3 decode vs exact fp32 decode: max |d| {err:.2e}")
    assert err < 2e-5
    rec = m.decode(t(g["ids"]).long().to(DEV))
    torch.testing.assert_close(rec.cpu(), t(g["rec"]), atol=LOGIT_TOL, rtol=0)
