"""GPU: the single-pass f16 precision mode (`set_precision('f16')`, MAGE_F16 in include/mage_hip.h) -- the bf16 mode's kernels, schedules and
data flow with IEEE half operands (v_mfma_f32_16x16x32_f16: same rate, 11 significand bits instead of 8).

What is checked, against the REFERENCE's goldens (tests/golden/, generated from /root/reference by tools/gen_golden.py) and against this
repo's other modes:
  * teacher-forced logits within F16_LOGIT_TOL of the reference's per-step logits (bf16's gate is 0.06), argmax == the reference's token
    wherever the reference decides by more than twice the measured error;
  * the invariants every mode keeps: incremental loop == full loop bitwise, B = 1 == a row of a batch bitwise, determinism, the fp32-stream
    form (`stream_bf16 = False`: fp32 residual stream + f16 copy) as well as the 16-bit stream;
  * BASELINE cfg2 size (B = 64, L = 16) through the size-independent properties;
  * loud refusals: training in 'f16', GEMM forms the f16 instantiations do not carry.
"""
import pytest
import torch

from mage_amd.utils import synth
from tests.helpers import build_mage, golden, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

F16_LOGIT_TOL = 0.012     # f16 operands (11 significand bits) through 6 blocks, fp32 accumulation: measured 0.003-0.005 on logits of
                          # magnitude ~2 (bf16: 0.02-0.03 under a 0.06 gate); the gate leaves 2x


def dev_batch(b):
    return {k: v.to(DEV) for k, v in b.items()}


def _teacher_forced_on_tokens(m, db, tok0, gen_tokens):
    """Decoder logits with the REFERENCE's generated tokens as context (tests/test_gpu_parity.py)."""
    B, L = tok0.shape[0], m.frames_length
    ctx = torch.cat([tok0.reshape(B, 1, -1), gen_tokens.reshape(B, L - 1, -1)[:, :L - 2]], 1).contiguous()
    dt = m._dt()
    ma = m._motion_anchor(tok0.reshape(B, -1).contiguous(), db, db.get("video_noise"))
    feats = m._frame_features(ctx, dt)
    lg = m.generate_model._run(ma if dt == torch.float32 else ma.to(dt), feats, B=B, hh=16, ww=16)
    return lg.view(B, L - 1, 16, 16, -1)


@pytest.mark.parametrize("tag,sub", [("mage_mnist_L16", (8, 8, 4)), ("mage_mnist_L6_ragged", (4, 4, 1))])
def test_f16_mode_against_reference_goldens(tag, sub):
    g = golden(tag)
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    kw = dict(digits=int(g["digits"]), text_len=int(g["text_len"]), ragged_text=bool(g["ragged"])) if "digits" in g.files else {}
    db = dev_batch(synth.synth_batch_mnist(B, L, seed=seed, **kw))
    want = t(g["gen_tokens"]).long().to(DEV)
    margin = t(g["margin"]).to(DEV)
    ref_sub = t(g["step_logits_sub"]).to(DEV)
    errs = {}
    for prec in ("bf16", "f16"):
        m = build_mage(synth.mnist_model_config(frames_length=L), seed, DEV).set_precision(prec)
        tok0 = m.first_stage_encode(db["images"][:, 0:1])[:, 0]
        lg = _teacher_forced_on_tokens(m, db, tok0, want)
        errs[prec] = (lg[:, :, ::sub[0], ::sub[1], ::sub[2]] - ref_sub).abs().max().item()
        if prec == "f16":
            am = lg.argmax(-1)
            hard = (am != want) & (margin > 2 * max(errs[prec], 1e-4))
            agree_tf = (am == want).float().mean().item()
    print(f"{tag}: teacher-forced max|d logit| vs the reference: bf16 {errs['bf16']:.4f}, f16 {errs['f16']:.4f} "
          f"({errs['bf16'] / max(errs['f16'], 1e-9):.1f}x smaller); f16 argmax == reference tokens {agree_tf:.4f}, "
          f"mismatches above 2x error margin: {int(hard.sum())}")
    assert errs["f16"] < F16_LOGIT_TOL and errs["f16"] < 0.5 * errs["bf16"] and int(hard.sum()) == 0 and agree_tf > 0.99
    m.autoregressive_generate(db)
    got = m.last_tokens
    bad_first = 0
    for b in range(B):
        diff = (got[b] != want[b]).flatten(1).any(1)
        if diff.any():
            f = int(diff.nonzero()[0])
            mism = got[b, f] != want[b, f]
            bad_first += int((mism & (margin[b, f] > F16_LOGIT_TOL)).sum())
    print(f"{tag}: f16 free-running token agreement with the reference {(got == want).float().mean().item():.4f}")
    assert bad_first == 0, "a clip's first divergence from the reference is at a position the reference decides by more than the f16 error"


@pytest.mark.parametrize("stream16", [True, False])
def test_f16_invariants_small(stream16):
    """incremental == full bitwise, B = 1 == row 0 of a batch, determinism; both residual-stream forms; logits track the fp32 mode."""
    m = build_mage(synth.mnist_model_config(frames_length=6), 5, DEV)
    batch = dev_batch(synth.synth_batch_mnist(4, 6, seed=5))
    tok32, lg32 = m.teacher_forced_logits(batch)
    m.set_precision("f16")
    m.generate_model.stream_bf16 = stream16
    tok16, lg16 = m.teacher_forced_logits(batch)
    assert torch.equal(tok32, tok16)                           # encoder + quantiser stay fp32-class: identical tokens
    err = (lg16 - lg32).abs()
    print(f"stream16={stream16}: f16 vs fp32 logits max |d| {err.max().item():.5f}, mean |d| {err.mean().item():.6f}")
    assert err.max().item() < 0.03 and err.mean().item() < 0.003
    v_full = m.autoregressive_generate(batch)
    t_full = m.last_tokens.clone()
    assert torch.isfinite(v_full).all() and v_full.abs().max().item() <= 1.0
    v_again = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, t_full) and torch.equal(v_again, v_full)
    m.ar_mode = "incremental"
    v_inc = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, t_full) and torch.equal(v_inc, v_full)
    one = {k: v[:1] for k, v in batch.items()}
    v1 = m.autoregressive_generate(one)
    assert torch.equal(m.last_tokens, t_full[:1]) and torch.equal(v1, v_full[:1])


@pytest.mark.parametrize("ar_mode", ["full", "incremental"])
def test_f16_single_clip_equals_row_of_a_batch_L16(ar_mode):
    L = 16
    m = build_mage(synth.mnist_model_config(frames_length=L), 3, DEV).set_precision("f16")
    m.ar_mode = ar_mode
    batch = dev_batch(synth.synth_batch_mnist(4, L, seed=9))
    v4 = m.autoregressive_generate(batch)
    t4 = m.last_tokens.clone()
    for r in (0, 3):
        one = {k: v[r:r + 1] for k, v in batch.items()}
        v1 = m.autoregressive_generate(one)
        assert torch.equal(m.last_tokens, t4[r:r + 1]) and torch.equal(v1, v4[r:r + 1])


def test_f16_full_size_cfg2():
    """BASELINE cfg2 size: determinism, shard == slice, incremental == full loop, first frame passed through."""
    m = build_mage(synth.mnist_model_config(frames_length=16), 0, DEV).set_precision("f16")
    batch = dev_batch(synth.synth_batch_mnist(64, 16, seed=3))
    v1 = m.autoregressive_generate(batch)
    tok1 = m.last_tokens.clone()
    v2 = m.autoregressive_generate(batch)
    assert torch.equal(tok1, m.last_tokens) and torch.equal(v1, v2)
    half = {k: v[32:] for k, v in batch.items()}
    vh = m.autoregressive_generate(half)
    assert torch.equal(m.last_tokens, tok1[32:]) and torch.equal(vh, v1[32:])
    assert torch.equal(v1[:, 0], batch["images"][:, 0]) and v1.abs().max().item() <= 1.0
    m.ar_mode = "incremental"
    vi = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, tok1) and torch.equal(vi, v1)


def test_f16_cater_config_runs_and_keeps_the_invariants():
    """config/mage_caterv1.yaml's model family (f8 VQ-VAE, randomness branch) at a small size: f16 tokens of the incremental loop == full loop."""
    L, B = 6, 2
    m = build_mage(synth.cater_model_config(frames_length=L), 0, DEV).set_precision("f16")
    cb = synth.synth_batch_cater(B, L, seed=2)
    cb["video_noise"] = torch.randn(B, 64, 16, 16, generator=torch.Generator().manual_seed(5))
    batch = dev_batch(cb)
    v = m.autoregressive_generate(batch)
    tk = m.last_tokens.clone()
    m.ar_mode = "incremental"
    vi = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, tk) and torch.equal(vi, v) and torch.isfinite(v).all()


def test_f16_refusals_are_loud():
    from mage_amd import ops as o
    m = build_mage(synth.mnist_model_config(frames_length=4), 1, DEV).set_precision("f16").train()
    for p in m.parameters():
        p.requires_grad_(True)
    m.first_stage_model.requires_grad_(False)
    with pytest.raises(ValueError, match="generation mode"):
        m(dev_batch(synth.synth_batch_mnist(2, 4, seed=1)))
    a = torch.randn(256, 64, device=DEV).half()
    w = torch.randn(256, 64, device=DEV).half()
    y = torch.empty(256, 256, device=DEV, dtype=torch.float16)
    with pytest.raises(Exception):                             # ReLU epilogue: a VQ-VAE form, bf16 / fp32 only
        o.gemm(a, w, y, M=256, N=256, K=64, lda=64, ldy=256, act=o.ACT_RELU)
    with pytest.raises(Exception):                             # bf16 output from f16 operands
        o.gemm(a, w, torch.empty(256, 256, device=DEV, dtype=torch.bfloat16), M=256, N=256, K=64, lda=64, ldy=256)
    with pytest.raises(Exception):                             # bf16 residual under f16 operands
        o.gemm(a, w, y, M=256, N=256, K=64, lda=64, ldy=256, residual=torch.zeros(256, 256, device=DEV, dtype=torch.bfloat16), ldr=256)


def test_f16_on_the_latent_first_stage_path():
    """set_precision('f16') with use_cids=False (MAGE+, BASELINE cfg5's MAGE side; refused until round 6): predicted latents against the reference's
    golden inside the f16 gate (bf16: 8x looser), closer to it than bf16, deterministic; and the cfg5-size call runs."""
    from tests.helpers import golden, t
    g = golden("mage_plus_small")
    B, L = int(g["B"]), int(g["L"])
    m = build_mage(synth.magep_model_config(frames_length=L, width=64, layers=3), int(g["seed"]), DEV)
    m.ma_encoder.mage_plus = False          # this fixture is the reference AS SHIPPED (mage_model.py:92 active)
    batch = dev_batch(synth.synth_batch_cater(B, L, seed=int(g["seed"]), text_len=int(g["text_len"]), vocab=50))
    batch["video_noise"] = t(g["noise"]).to(DEV)
    errs = {}
    for prec in ("bf16", "f16"):
        m.set_precision(prec)
        v = m.autoregressive_generate(batch)
        lat = m.last_logits.clone()
        errs[prec] = (lat.cpu() - t(g["pred_latents"])).abs().max().item()
        v2 = m.autoregressive_generate(batch)
        assert torch.equal(v, v2) and torch.equal(lat, m.last_logits) and torch.isfinite(v).all()
    print(f"MAGE+ predicted latents vs the reference's golden: bf16 {errs['bf16']:.2e}, f16 {errs['f16']:.2e}")
    # (free-running in latent space: every iteration feeds its prediction back, so the error compounds over the L - 1 iterations)
    assert errs["f16"] < 2e-2 and errs["f16"] < 0.5 * errs["bf16"], errs
    big = build_mage(synth.magep_model_config(frames_length=8), 0, DEV).set_precision("f16")
    bb = dev_batch(synth.synth_batch_cater(4, 8, seed=2, vocab=50, text_len=24))
    bb["video_noise"] = torch.randn(4, 64, 16, 16, generator=torch.Generator().manual_seed(6)).to(DEV)
    vv = big.autoregressive_generate(bb)
    assert tuple(vv.shape) == (4, 8, 3, 128, 128) and torch.isfinite(vv).all() and torch.isfinite(big.last_logits).all()
