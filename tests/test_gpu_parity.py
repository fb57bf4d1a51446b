"""GPU parity tests proper: the HIP product path (mage_amd.modules, through the C ABI) against
(a) the committed golden vectors produced by the reference itself and (b) the CPU oracle on the same
seeded inputs.  Bar: bit-exact token indices wherever the reference's own top-2 margin is above fp32
rounding noise, fp32 logits / frames within 1e-4 (north_star).  bf16 mode is checked against
bf16-rounding tolerances and reported, and full-size (BASELINE cfg2) runs are checked through
size-independent properties (shard invariance, determinism, causality of the AR loop)."""

import numpy as np
import pytest
import torch

from mage_amd import config
from mage_amd.utils import synth
from oracle import mage_oracle as O
from tests.helpers import assert_tokens, build_mage, build_vqvae, chk, cpu_sd, golden, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOK_TOL = 2e-5          # index parity is asserted wherever the reference's top-2 margin exceeds this
LOGIT_TOL = 1e-4        # north_star: fp32 logits within 1e-4


def dev_batch(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


def test_product_path_loaded_native_library():
    from mage_amd import _lib
    l = _lib.lib(0)
    assert l.mage_abi_version() == _lib.ABI_VERSION
    with open("/proc/self/maps") as f:
        assert "libmage_hip.so" in f.read()


def test_vqvae_f4_golden_tokens_and_frames():
    g = golden("vqvae_f4")
    m = build_vqvae(1, 4, 256, 512, int(g["seed"]), DEV)
    x = t(g["x"]).to(DEV)
    z = m._encode_features(x).view(4, 16, 16, 256).permute(0, 3, 1, 2)
    torch.testing.assert_close(z[:, :8].cpu(), t(g["z_e_slice"]), atol=5e-5, rtol=1e-5)
    ids = m.encode(x)
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (4, 16, 16)
    n_soft = assert_tokens(ids.cpu(), g["ids"], g["margin"], TOK_TOL, "f4 encode")
    assert n_soft == 0, f"{n_soft} near-tie flips"
    rec = m.decode(t(g["ids"]).long().to(DEV))
    torch.testing.assert_close(rec.cpu(), t(g["rec"]), atol=LOGIT_TOL, rtol=0)
    x_tilde, z_e, z_q = m(x)
    np.testing.assert_allclose(chk(x_tilde.cpu()), g["fwd_x_tilde_chk"], rtol=1e-4)
    np.testing.assert_allclose(chk(z_q.cpu()), g["z_q_chk"], rtol=1e-5)
    assert tuple(z_e.shape) == (4, 256, 16, 16)
    # bf16 decode mode: same tokens in, frames within bf16-class error
    m.set_precision("bf16")
    rec16 = m.decode(t(g["ids"]).long().to(DEV))
    assert (rec16.cpu() - t(g["rec"])).abs().max().item() < 5e-2
    # the last transposed convolution's taps are taken on the sub-pixel GEMMs' tiles (mage_gemm_desc::head_w): the same bf16 rows in another
    # summation order than the separate head GEMM
    with config.override(decode_head_fusion=False):
        rec16_unfused = m.decode(t(g["ids"]).long().to(DEV))
    with config.lib_option("gemm_no_taps8", 1):                 # the padded-taps kernel switched off in the library: the decode falls back, no error
        rec16_generic = m.decode(t(g["ids"]).long().to(DEV))
    assert (rec16 - rec16_generic).abs().max().item() < 2e-2
    assert (rec16 - rec16_unfused).abs().max().item() < 5e-6
    with config.override(decode_phase_merge=False):              # the four sub-pixel launches instead of one: the same bits
        rec16_four = m.decode(t(g["ids"]).long().to(DEV))
    assert torch.equal(rec16, rec16_four)
    # the first ResBlock in one launch (mage_resblock_table) against embedding + table sum + 1x1 GEMM: the same bits
    with config.override(decode_resblock_fusion=False):
        rec16_three = m.decode(t(g["ids"]).long().to(DEV))
    assert torch.equal(rec16, rec16_three)


def test_vqvae_f8_golden_tokens_and_frames():
    g = golden("vqvae_f8")
    m = build_vqvae(3, 8, int(g["dim"]), int(g["K"]), int(g["seed"]), DEV)
    x = synth.synth_batch_cater(2, 1, seed=int(g["seed"]))["images"][:, 0].contiguous().to(DEV)
    z = m._encode_features(x).view(2, 16, 16, -1).permute(0, 3, 1, 2)
    torch.testing.assert_close(z[:, :8].cpu(), t(g["z_e_slice"]), atol=5e-5, rtol=1e-5)
    ids = m.encode(x)
    assert assert_tokens(ids.cpu(), g["ids"], g["margin"], TOK_TOL, "f8 encode") == 0
    rec = m.decode(t(g["ids"]).long().to(DEV))
    torch.testing.assert_close(rec[..., ::4, ::4].cpu(), t(g["rec_sub"]), atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(chk(rec.cpu()), g["rec_chk"], rtol=1e-4)


def test_vqvae_f8_bf16_decode_fusions_keep_the_frames():
    """The f8 decoder's bf16 path at a batch every fusion engages at (256 frames: the 256 x 64 tile list fills the chip in every block):
    a block's leading ReLU on the operand fragments of its first 1x1 convolution (mage_gemm_desc::a_relu) gives the bits of the separate
    ReLU pass; the last block's closing convolution + identity path + ReLU + RGB head in one launch (head_w with a residual, four sums per
    row) gives the unfused frames to bf16 summation-order error; both stay within bf16-class error of the exact decode."""
    g = golden("vqvae_f8")
    m = build_vqvae(3, 8, int(g["dim"]), int(g["K"]), int(g["seed"]), DEV)
    ids = torch.randint(0, int(g["K"]), (256, 16, 16), generator=torch.Generator().manual_seed(11)).to(DEV)
    ids[:2] = t(g["ids"]).long().to(DEV)
    rec32 = m.decode(ids[:8])
    torch.testing.assert_close(rec32[:2, :, ::4, ::4].cpu(), t(g["rec_sub"]), atol=LOGIT_TOL, rtol=0)
    m.set_precision("bf16")
    rec = m.decode(ids)
    assert torch.isfinite(rec).all() and (rec[:8] - rec32).abs().max().item() < 5e-2
    with config.override(decode_relu_fold=False):
        rec_relu_pass = m.decode(ids)
    assert torch.equal(rec, rec_relu_pass)
    with config.override(decode_head_fusion=False):
        rec_unfused = m.decode(ids)
    print(f"f8 bf16 decode: fused tail vs separate launches max |d| {(rec - rec_unfused).abs().max().item():.2e}, "
          f"vs exact decode {(rec[:8] - rec32).abs().max().item():.2e}")
    assert (rec - rec_unfused).abs().max().item() < 2e-3
    assert torch.equal(m.decode(ids[:8]), rec[:8])              # few frames (no fold: the tile list would not fill the chip): the same frames


@pytest.mark.parametrize("tag", ["mage_mnist_L4", "mage_mnist_L6_ragged"])
def test_mage_stages_against_reference_goldens(tag):
    g = golden(tag)
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    m = build_mage(synth.mnist_model_config(frames_length=L), seed, DEV)
    batch = synth.synth_batch_mnist(B, L, seed=seed, digits=int(g["digits"]), text_len=int(g["text_len"]), ragged_text=bool(g["ragged"]))
    db = dev_batch(batch)
    txt = m.text_encoder(db["text"])
    torch.testing.assert_close(txt.cpu(), t(g["text_emb"]), atol=5e-5, rtol=1e-5)       # padded rows included
    tok0 = m.first_stage_encode(db["images"][:, 0:1])[:, 0]
    assert torch.equal(tok0.cpu(), t(g["tok0"]).long())
    ma = m._motion_anchor(tok0.reshape(B, -1), db, None).view(B, 16, 16, -1)
    torch.testing.assert_close(ma[:, ::4, ::4].cpu(), t(g["motion_sub"]), atol=LOGIT_TOL, rtol=1e-5)
    video = m.autoregressive_generate(db)
    assert tuple(video.shape) == (B, L, 1, 64, 64) and video.dtype == torch.float32
    n_soft = assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "AR tokens")
    assert n_soft == 0, f"{n_soft} argmax flips inside fp32 rounding noise"
    torch.testing.assert_close(video.cpu(), t(g["video"]), atol=LOGIT_TOL, rtol=0)
    # teacher-forced logits of the final iteration vs the reference's per-step logits (identical by causality)
    torch.testing.assert_close(m.last_logits[:, :, ::4, ::4].cpu(), t(g["step_logits_sub"]), atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(chk(m.last_logits.cpu()), g["step_logits_chk"], rtol=1e-4)
    loss, ld = m(db)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    assert set(ld.keys()) == set(g["loss_dict_keys"].tolist())


def test_mage_reduced_width_against_golden_and_oracle():
    g = golden("mage_small_d64")
    cfg = synth.mnist_model_config(frames_length=int(g["L"]), width=64, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, int(g["seed"]), DEV)
    batch = synth.synth_batch_mnist(int(g["B"]), int(g["L"]), seed=int(g["seed"]), text_len=int(g["text_len"]), ragged_text=True)
    video = m.autoregressive_generate(dev_batch(batch))
    assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "AR tokens d64") == 0
    torch.testing.assert_close(m.last_logits.cpu(), t(g["step_logits"]), atol=LOGIT_TOL, rtol=0)
    torch.testing.assert_close(video.cpu(), t(g["video"]), atol=LOGIT_TOL, rtol=0)
    # the oracle on the same weights agrees too (oracle == reference is pinned on CPU)
    ov = O.mage_generate(cpu_sd(m), batch, int(g["L"]))
    torch.testing.assert_close(video.cpu(), ov, atol=LOGIT_TOL, rtol=0)


def test_mage_cater_randomness_branch_golden():
    g = golden("mage_cater_small")
    cfg = synth.cater_model_config(frames_length=int(g["L"]), width=64, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, int(g["seed"]), DEV)
    batch = synth.synth_batch_cater(int(g["B"]), int(g["L"]), seed=int(g["seed"]), text_len=int(g["text_len"]))
    db = dev_batch(batch)
    db["video_noise"] = t(g["noise"]).to(DEV)
    video = m.autoregressive_generate(db)
    assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "AR tokens cater") == 0
    torch.testing.assert_close(m.last_logits.cpu(), t(g["step_logits"]), atol=LOGIT_TOL, rtol=0)      # measured 2.9e-6 (profiles/r02_parity_report.txt)
    torch.testing.assert_close(video[..., ::4, ::4].cpu(), t(g["video_sub"]), atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(chk(video.cpu()), g["video_chk"], rtol=1e-4)


def test_mage_cater_forward_randomness_golden():
    """MAGE.forward with randomness=True through the HIP path (Conv3d video prior as temporal-tap GEMMs + GroupNorm kernels,
    reparameterisation with the reference's noise injected, KL + speed-l2 terms) against the reference's own loss values, its
    conv3d output and its teacher-forced logits (fp32 mode), and against the CPU oracle."""
    from oracle import mage_oracle as O
    from tests.helpers import cpu_sd
    g = golden("mage_cater_forward_small")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    cfg = synth.cater_model_config(frames_length=L, width=int(g["width"]), layers=int(g["layers"]), vq_dim=int(g["vq_dim"]), K=int(g["K"]))
    m = build_mage(cfg, seed, DEV)
    batch = synth.synth_batch_cater(B, L, seed=seed, text_len=int(g["text_len"]))
    db = dev_batch(batch)
    db["reparam_noise"] = t(g["eps"]).to(DEV)
    extras = {}
    tok, logits = m.teacher_forced_logits(db, extras)
    R = 16
    prior = extras["prior"].view(B, R, R, -1).permute(0, 3, 1, 2).cpu()                     # rows -> NCHW
    torch.testing.assert_close(prior[:, ::4, ::2, ::2], t(g["prior_sub"]), atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(chk(prior), g["prior_chk"], rtol=1e-4)
    torch.testing.assert_close(logits[:, ::3, ::4, ::4, ::8].cpu(), t(g["logits_sub"]), atol=LOGIT_TOL, rtol=1e-4)
    loss, ld = m(db)
    assert abs(ld["val/prediction"] - float(g["prediction"])) < 1e-4
    assert abs(ld["val/kl_loss"] - float(g["kl_loss"])) < 1e-4 * max(1.0, abs(float(g["kl_loss"])))
    assert abs(loss.item() - float(g["final_loss"])) < 1e-4 * max(1.0, abs(float(g["final_loss"])))
    # and the oracle on the same inputs (what smoke()/bench compare against on a box without the reference)
    sd = cpu_sd(m)
    final, parts, _, _ = O.mage_forward_loss_random(sd, batch, L, t(g["eps"]), alpha=cfg["params"]["alpha"], beta=cfg["params"]["beta"])
    assert abs(loss.item() - final.item()) < 1e-4 * max(1.0, abs(final.item()))


def test_mage_forward_test_flag_replaces_the_video_embedding_by_noise():
    """MAGE.forward(batch, test_flag=True) (mage_model.py:604-605): the reparameterised embedding is computed (its mu / logvar feed the
    KL term) and then replaced by noise; against the oracle with the same noise injected."""
    from oracle import mage_oracle as O
    from tests.helpers import cpu_sd
    B, L, seed = 2, 10, 71
    cfg = synth.cater_model_config(frames_length=L, width=64, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, seed, DEV)
    batch = synth.synth_batch_cater(B, L, seed=seed, text_len=9)
    g = torch.Generator().manual_seed(seed)
    eps, noise = torch.randn(B, 64, 16, 16, generator=g), torch.randn(B, 64, 16, 16, generator=g)
    db = dev_batch(batch)
    db["reparam_noise"], db["video_noise"] = eps.to(DEV), noise.to(DEV)
    with torch.no_grad():
        loss, ld = m(db, test_flag=True)
        loss_plain, _ = m(db)
    final, parts, _, _ = O.mage_forward_loss_random(cpu_sd(m), batch, L, eps, alpha=cfg["params"]["alpha"], beta=cfg["params"]["beta"],
                                                    test_noise=noise)
    assert abs(loss.item() - final.item()) < 1e-4 * max(1.0, abs(final.item()))
    assert abs(ld["val/kl_loss"] - parts["kl_loss"]) < 1e-4 * max(1.0, abs(parts["kl_loss"]))
    assert abs(ld["val/prediction"] - parts["prediction"]) < 1e-4 and abs(loss.item() - loss_plain.item()) > 1e-6


def test_mage_plus_forward_latent_golden():
    """MAGE.forward for use_cids=False on the HIP path (Linear embedding written straight into the padded frame buffer of the
    video prior, MSE kernel, PID-controlled beta) against the reference's own loss values and predicted latents."""
    g = golden("mage_plus_forward_small")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    cfg = synth.magep_model_config(frames_length=L, width=int(g["width"]), layers=int(g["layers"]))
    m = build_mage(cfg, seed, DEV)
    m.ma_encoder.mage_plus = False          # this fixture is the reference AS SHIPPED (mage_model.py:92 active)
    db = dev_batch(synth.synth_batch_cater(B, L, seed=seed, text_len=int(g["text_len"]), vocab=50))
    db["reparam_noise"] = t(g["eps"]).to(DEV)
    loss, ld = m(db)
    pred = m.last_logits.view(B, L - 1, 16, 16, -1)[..., :4].cpu()
    torch.testing.assert_close(pred, t(g["pred"]), atol=LOGIT_TOL, rtol=0)
    assert abs(ld["val/prediction"] - float(g["prediction"])) < 1e-4 * max(1.0, abs(float(g["prediction"])))
    assert abs(ld["val/kl_loss"] - float(g["kl_loss"])) < 1e-4 * max(1.0, abs(float(g["kl_loss"])))
    assert abs(ld["val/beta"] - float(g["beta"])) < 1e-6
    assert abs(loss.item() - float(g["final_loss"])) < 1e-4 * max(1.0, abs(float(g["final_loss"])))


def test_mage_L16_golden_tokens():
    """BASELINE cfg1 model (MNIST f4, L=16) at B=2: the reference's own token sequence."""
    g = golden("mage_mnist_L16")
    m = build_mage(synth.mnist_model_config(frames_length=16), int(g["seed"]), DEV)
    batch = synth.synth_batch_mnist(int(g["B"]), 16, seed=int(g["seed"]))
    video = m.autoregressive_generate(dev_batch(batch))
    n_soft = assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "AR tokens L16")
    assert n_soft == 0
    torch.testing.assert_close(m.last_logits[:, :, ::8, ::8, ::4].cpu(), t(g["step_logits_sub"]), atol=LOGIT_TOL, rtol=0)
    torch.testing.assert_close(video[:, :, :, ::2, ::2].cpu(), t(g["video_sub"]), atol=LOGIT_TOL, rtol=0)


def test_bf16_mode_tracks_fp32_teacher_forced():
    """bf16-MFMA performance mode: teacher-forced logits against the fp32 mode of the same kernels.
    Reported, and gated at a bf16-class tolerance (the 1e-4 gate is the fp32 mode's)."""
    m = build_mage(synth.mnist_model_config(frames_length=6), 5, DEV)
    batch = dev_batch(synth.synth_batch_mnist(4, 6, seed=5))
    tok32, lg32 = m.teacher_forced_logits(batch)
    m.set_precision("bf16")
    tok16, lg16 = m.teacher_forced_logits(batch)
    assert torch.equal(tok32, tok16)                           # encoder + quantiser stay fp32: identical tokens
    err = (lg16 - lg32).abs()
    agree = (lg16.argmax(-1) == lg32.argmax(-1)).float().mean().item()
    print(f"bf16 vs fp32 logits: max |d| {err.max().item():.4f}, mean |d| {err.mean().item():.5f}, argmax agreement {agree:.4f}")
    assert err.max().item() < 0.25 and err.mean().item() < 0.02 and agree > 0.9
    v = m.autoregressive_generate(batch)
    assert torch.isfinite(v).all() and v.abs().max().item() <= 1.0


def test_full_size_properties_cfg2_shape():
    """BASELINE cfg2 sizes (B=64, L=16, d=512) through size-independent properties:
    determinism and batch-shard invariance (the multi-GPU contract: clips are independent)."""
    m = build_mage(synth.mnist_model_config(frames_length=16), 0, DEV).set_precision("bf16")
    batch = dev_batch(synth.synth_batch_mnist(64, 16, seed=3))
    v1 = m.autoregressive_generate(batch)
    tok1 = m.last_tokens.clone()
    v2 = m.autoregressive_generate(batch)
    assert torch.equal(tok1, m.last_tokens) and torch.equal(v1, v2)                 # bitwise deterministic
    half = {k: v[32:] for k, v in batch.items()}
    vh = m.autoregressive_generate(half)
    assert torch.equal(m.last_tokens, tok1[32:]) and torch.equal(vh, v1[32:])       # shard == slice of the whole
    assert torch.equal(v1[:, 0], batch["images"][:, 0])                              # first frame is passed through
    assert v1.abs().max().item() <= 1.0


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_incremental_decoding_is_bit_identical_to_reference_loop(precision):
    """temporal-KV-cache decoding processes each position once; same kernels, same per-row arithmetic, so
    the token sequence and the frames are bitwise those of the reference's full-recompute loop -- and the golden ones."""
    g = golden("mage_mnist_L6_ragged")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    m = build_mage(synth.mnist_model_config(frames_length=L), seed, DEV).set_precision(precision)
    batch = dev_batch(synth.synth_batch_mnist(B, L, seed=seed, digits=int(g["digits"]), text_len=int(g["text_len"]), ragged_text=True))
    v_full = m.autoregressive_generate(batch)
    t_full = m.last_tokens.clone()
    m.ar_mode = "incremental"
    v_inc = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, t_full) and torch.equal(v_inc, v_full)
    if precision == "fp32":
        assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "incremental AR tokens") == 0
        torch.testing.assert_close(v_inc.cpu(), t(g["video"]), atol=LOGIT_TOL, rtol=0)


@pytest.mark.parametrize("stream_bf16", [True, False])
def test_bf16_residual_stream_forms(stream_bf16):
    """bf16 mode keeps x in bf16 between the blocks (default) or as the fp32 stream + bf16 copy (`stream_bf16 = False`): for both, the
    incremental loop == the full loop bitwise, B = 1 == row 0 of a batch, and the teacher-forced logits stay inside the bf16 gate."""
    m = build_mage(synth.mnist_model_config(frames_length=6), 5, DEV)
    batch = dev_batch(synth.synth_batch_mnist(4, 6, seed=5))
    _, lg32 = m.teacher_forced_logits(batch)
    m.set_precision("bf16")
    m.generate_model.stream_bf16 = stream_bf16
    _, lg16 = m.teacher_forced_logits(batch)
    err = (lg16 - lg32).abs()
    print(f"stream_bf16={stream_bf16}: bf16 vs fp32 logits max |d| {err.max().item():.4f}, mean |d| {err.mean().item():.5f}")
    assert err.max().item() < 0.25 and err.mean().item() < 0.02
    v_full = m.autoregressive_generate(batch)
    t_full = m.last_tokens.clone()
    m.ar_mode = "incremental"
    v_inc = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, t_full) and torch.equal(v_inc, v_full)
    one = {k: v[:1] for k, v in batch.items()}
    v1 = m.autoregressive_generate(one)
    assert torch.equal(m.last_tokens, t_full[:1]) and torch.equal(v1, v_full[:1])


def test_incremental_decoding_full_size_cfg2():
    m = build_mage(synth.mnist_model_config(frames_length=16), 0, DEV).set_precision("bf16")
    batch = dev_batch(synth.synth_batch_mnist(64, 16, seed=3))
    v_full = m.autoregressive_generate(batch)
    t_full = m.last_tokens.clone()
    m.ar_mode = "incremental"
    v_inc = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, t_full) and torch.equal(v_inc, v_full)


def test_multistream_clip_groups_are_bit_identical():
    m = build_mage(synth.mnist_model_config(frames_length=6), 5, DEV).set_precision("bf16")
    batch = dev_batch(synth.synth_batch_mnist(8, 6, seed=5))
    v1 = m.autoregressive_generate(batch)
    t1 = m.last_tokens.clone()
    m.streams = 2
    v2 = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, t1) and torch.equal(v1, v2)
    m.streams, m.ar_mode = 4, "incremental"
    v3 = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, t1) and torch.equal(v1, v3)
    # the fast parity mode on side streams: its split-precision weight copies are built on the caller's stream first (_warm_derived)
    m2 = build_mage(synth.mnist_model_config(frames_length=6), 5, DEV).set_precision("f16x3")
    m2.streams = 2
    v4 = m2.autoregressive_generate(batch)
    t4 = m2.last_tokens.clone()
    m2.streams = 1
    v5 = m2.autoregressive_generate(batch)
    assert torch.equal(m2.last_tokens, t4) and torch.equal(v4, v5)


def test_mage_plus_latent_path_golden():
    """BASELINE cfg5, MAGE side only (use_cids=False over a latent first stage): golden from the reference's MAGE driven
    with the same stand-in first stage (tests/standin_first_stage.py; the real `ldm` AutoencoderKL is outside the mount)."""
    g = golden("mage_plus_small")
    B, L = int(g["B"]), int(g["L"])
    m = build_mage(synth.magep_model_config(frames_length=L, width=64, layers=3), int(g["seed"]), DEV)
    m.ma_encoder.mage_plus = False          # this fixture is the reference AS SHIPPED (mage_model.py:92 active)
    batch = dev_batch(synth.synth_batch_cater(B, L, seed=int(g["seed"]), text_len=int(g["text_len"]), vocab=50))
    batch["video_noise"] = t(g["noise"]).to(DEV)
    video = m.autoregressive_generate(batch)
    assert tuple(video.shape) == (B, L, 3, 128, 128)
    torch.testing.assert_close(m.last_logits.cpu(), t(g["pred_latents"]), atol=LOGIT_TOL, rtol=0)       # measured 8.4e-6
    torch.testing.assert_close(video[..., ::4, ::4].cpu(), t(g["video_sub"]), atol=LOGIT_TOL, rtol=0)    # measured 1.0e-5


def test_mage_cater_fullwidth_golden():
    """cfg4's model at FULL width (config/mage_caterv1.yaml: d=512, 6 blocks, f8 VQ-VAE dim 256 -> codebook D=1024, K=512,
    randomness branch with the noise injected) on a short clip, fp32 mode: the kernel dispatch of the full-size CATER runs
    (256-wide tiles, K=1024 quantiser, f8 stack at dim 256) against the reference's own tokens / logits / frames at 1e-4."""
    g = golden("mage_cater_fullwidth")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    m = build_mage(synth.cater_model_config(frames_length=L), seed, DEV)
    db = dev_batch(synth.synth_batch_cater(B, L, seed=seed, text_len=int(g["text_len"])))
    db["video_noise"] = t(g["noise"]).to(DEV)
    x0 = db["images"][:, 0].contiguous()
    z = m.first_stage_model._encode_features(x0).view(B, 16, 16, -1).permute(0, 3, 1, 2)
    torch.testing.assert_close(z[:, :8].cpu(), t(g["z_e_slice"]), atol=LOGIT_TOL, rtol=1e-5)
    tok0 = m.first_stage_encode(db["images"][:, 0:1])[:, 0]
    assert assert_tokens(tok0.cpu(), g["tok0"], g["tok0_margin"], TOK_TOL, "f8 encode, D=1024") == 0
    ma = m._motion_anchor(tok0.reshape(B, -1), db, db["video_noise"]).view(B, 16, 16, -1)
    torch.testing.assert_close(ma[:, ::4, ::4].cpu(), t(g["motion_sub"]), atol=LOGIT_TOL, rtol=1e-5)
    video = m.autoregressive_generate(db)
    assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "AR tokens cater full width") == 0
    torch.testing.assert_close(m.last_logits[:, :, ::4, ::4].cpu(), t(g["step_logits_sub"]), atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(chk(m.last_logits.cpu()), g["step_logits_chk"], rtol=1e-4)
    torch.testing.assert_close(video[..., ::4, ::4].cpu(), t(g["video_sub"]), atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(chk(video.cpu()), g["video_chk"], rtol=1e-4)
    # incremental decoding on this config: same tokens, same frames
    m.ar_mode = "incremental"
    v_inc = m.autoregressive_generate(db)
    assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "incremental AR tokens cater full width") == 0
    assert torch.equal(v_inc, video)


def test_mage_plus_transformer_block_variant_golden():
    """MAGE+ with the TransformerBlock variant of mage_model.py:93 (ln_q / ln_kv): MAGE(use_cids=False) turns it on by itself;
    sampling (motion anchor, predicted latents, frames) and the teacher-forced loss against the reference run with that line."""
    g = golden("mage_plus_block_small")
    B, L = int(g["B"]), int(g["L"])
    m = build_mage(synth.magep_model_config(frames_length=L, width=int(g["width"]), layers=int(g["layers"])), int(g["seed"]), DEV)
    assert m.ma_encoder.mage_plus is True
    batch = dev_batch(synth.synth_batch_cater(B, L, seed=int(g["seed"]), text_len=int(g["text_len"]), vocab=50))
    batch["video_noise"] = t(g["noise"]).to(DEV)
    video = m.autoregressive_generate(batch)
    torch.testing.assert_close(m.last_logits.cpu(), t(g["pred_latents"]), atol=LOGIT_TOL, rtol=0)
    torch.testing.assert_close(video[..., ::4, ::4].cpu(), t(g["video_sub"]), atol=LOGIT_TOL, rtol=0)
    Lf, fseed = int(g["fwd_L"]), int(g["fwd_seed"])
    mf = build_mage(synth.magep_model_config(frames_length=Lf, width=int(g["width"]), layers=int(g["layers"])), fseed, DEV)
    db = dev_batch(synth.synth_batch_cater(B, Lf, seed=fseed, text_len=int(g["text_len"]), vocab=50))
    db["reparam_noise"] = t(g["fwd_eps"]).to(DEV)
    loss, ld = mf(db)
    pred = mf.last_logits.view(B, Lf - 1, 16, 16, -1)[..., :4].cpu()
    torch.testing.assert_close(pred[:, ::3], t(g["fwd_pred_sub"]), atol=LOGIT_TOL, rtol=0)
    assert abs(ld["val/prediction"] - float(g["fwd_prediction"])) < 1e-4 * max(1.0, abs(float(g["fwd_prediction"])))
    assert abs(ld["val/kl_loss"] - float(g["fwd_kl_loss"])) < 1e-4 * max(1.0, abs(float(g["fwd_kl_loss"])))
    assert abs(ld["val/beta"] - float(g["fwd_beta"])) < 1e-6
    assert abs(loss.item() - float(g["fwd_final_loss"])) < 1e-4 * max(1.0, abs(float(g["fwd_final_loss"])))


# ------------------------------------------------------------------------------------------------ the benchmarked (bf16) mode
BF16_LOGIT_TOL = 0.06    # bf16 operands (8 mantissa bits) through 6 blocks with fp32 accumulation, fp32 residual stream: measured
                         # max |d logit| 0.02-0.03 on logits of magnitude ~2; the gate leaves 2x


def _teacher_forced_on_tokens(m, db, tok0, gen_tokens):
    """Decoder logits with the REFERENCE's generated tokens as context (slot 0 = frame 0's tokens, slot i = the reference's
    frame i): by causality these are the logits the reference's AR loop saw at each step."""
    B, L = tok0.shape[0], m.frames_length
    ctx = torch.cat([tok0.reshape(B, 1, -1), gen_tokens.reshape(B, L - 1, -1)[:, :L - 2]], 1).contiguous()
    dt = m._dt()
    ma = m._motion_anchor(tok0.reshape(B, -1).contiguous(), db, db.get("video_noise"))
    feats = m._frame_features(ctx, dt)
    lg = m.generate_model._run(ma if dt == torch.float32 else ma.to(dt), feats, B=B, hh=16, ww=16)
    return lg.view(B, L - 1, 16, 16, -1)


@pytest.mark.parametrize("tag,sub", [("mage_mnist_L16", (8, 8, 4)), ("mage_mnist_L6_ragged", (4, 4, 1))])
def test_bf16_mode_against_reference_goldens(tag, sub):
    """The BENCHMARKED precision against the reference itself (not against this repo's fp32 mode).
    (1) Teacher-forced on the reference's own token sequence, the bf16 logits are within BF16_LOGIT_TOL of the reference's
        per-step logits, and the bf16 argmax IS the reference's token wherever the reference's top-2 margin exceeds twice
        the measured bf16 error.
    (2) Free-running, a clip's tokens equal the reference's up to the first position whose reference margin is inside the
        bf16 error (after a flip the inputs differ, so later frames legitimately differ); the agreement rate is printed."""
    g = golden(tag)
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    m = build_mage(synth.mnist_model_config(frames_length=L), seed, DEV).set_precision("bf16")
    kw = dict(digits=int(g["digits"]), text_len=int(g["text_len"]), ragged_text=bool(g["ragged"])) if "digits" in g.files else {}
    db = dev_batch(synth.synth_batch_mnist(B, L, seed=seed, **kw))
    want = t(g["gen_tokens"]).long().to(DEV)                       # [B, L-1, 16, 16]
    margin = t(g["margin"]).to(DEV)
    tok0 = m.first_stage_encode(db["images"][:, 0:1])[:, 0]
    lg = _teacher_forced_on_tokens(m, db, tok0, want)
    ref_sub = t(g["step_logits_sub"]).to(DEV)
    got_sub = lg[:, :, ::sub[0], ::sub[1], ::sub[2]]
    err = (got_sub - ref_sub).abs().max().item()
    am = lg.argmax(-1)
    hard = (am != want) & (margin > 2 * max(err, 1e-3))
    agree_tf = (am == want).float().mean().item()
    print(f"{tag}: bf16 teacher-forced vs reference: max|d logit| {err:.4f}, argmax == reference tokens {agree_tf:.4f}, "
          f"mismatches above 2x error margin: {int(hard.sum())}")
    assert err < BF16_LOGIT_TOL and int(hard.sum()) == 0 and agree_tf > 0.97
    m.autoregressive_generate(db)
    got = m.last_tokens
    agree = (got == want).float().mean().item()
    bad_first = 0
    for b in range(B):
        diff = (got[b] != want[b]).flatten(1).any(1)                # per frame
        if diff.any():
            f = int(diff.nonzero()[0])
            mism = got[b, f] != want[b, f]
            bad_first += int((mism & (margin[b, f] > BF16_LOGIT_TOL)).sum())
    print(f"{tag}: bf16 free-running token agreement with the reference {agree:.4f}")
    assert bad_first == 0, "a clip's first divergence from the reference is at a position the reference decides by more than the bf16 error"


def test_full_size_properties_cfg3_double_mnist_per_gpu_share():
    """BASELINE cfg3 at its PER-GPU size (256 clips over 8 GPUs = 32 clips, L = 16, two moving digits, captions of 16 / 18 / 20 tokens
    right-padded to 20: the padded-text quirk of SURVEY 8a9/a10) through the size-independent properties: determinism,
    shard == slice of the whole batch, incremental == full loop, first frame passed through, range."""
    B, L = 32, 16
    m = build_mage(synth.mnist_model_config(frames_length=L), 0, DEV).set_precision("bf16")
    cb = synth.synth_batch_mnist(B, L, seed=11, digits=2, caption_lengths=(16, 18, 20))
    assert cb["text"].shape[1] == 20 and len({int(n) for n in (cb["text"] != 0).sum(1)}) > 1          # ragged, right-padded
    batch = dev_batch(cb)
    v1 = m.autoregressive_generate(batch)
    tok1 = m.last_tokens.clone()
    assert tuple(v1.shape) == (B, L, 1, 64, 64) and v1.abs().max().item() <= 1.0
    assert torch.equal(v1[:, 0], batch["images"][:, 0])
    v2 = m.autoregressive_generate(batch)
    assert torch.equal(tok1, m.last_tokens) and torch.equal(v1, v2)
    q = {k: v[8:16] for k, v in batch.items()}                                       # an interior shard: 8 clips
    vq = m.autoregressive_generate(q)
    assert torch.equal(m.last_tokens, tok1[8:16]) and torch.equal(vq, v1[8:16])
    m.ar_mode = "incremental"
    vi = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, tok1) and torch.equal(vi, v1)


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_batch_without_speed_key_against_the_oracle(precision):
    """The 'speed' key is optional (mage_model.py:611,666 guard it with `'speed' in batch`): sampling and the teacher-forced loss on
    a batch WITHOUT it, against the CPU oracle on the same weights -- and the result differs from the with-speed one."""
    L, B, seed = 4, 2, 31
    m = build_mage(synth.mnist_model_config(frames_length=L), seed, DEV).set_precision(precision)
    full = synth.synth_batch_mnist(B, L, seed=seed)
    batch = {k: v for k, v in full.items() if k != "speed"}
    video = m.autoregressive_generate(dev_batch(batch))
    want, want_tok, _, trace = O.mage_generate(cpu_sd(m), batch, L, return_trace=True)
    top2 = trace.topk(2, dim=-1)[0]
    margin = (top2[..., 0] - top2[..., 1]).abs()
    assert assert_tokens(m.last_tokens.cpu(), want_tok, margin, TOK_TOL, "AR tokens, no speed key") == 0
    torch.testing.assert_close(video.cpu(), want, atol=LOGIT_TOL, rtol=0)
    loss, _ = m(dev_batch(batch))
    want_loss = O.mage_forward_loss(cpu_sd(m), batch, L)[0]
    assert abs(loss.item() - float(want_loss)) < 1e-4
    loss_speed, _ = m(dev_batch(full))
    assert abs(loss_speed.item() - loss.item()) > 1e-6                              # the speed embedding is really applied when given


@pytest.mark.parametrize("precision", ["bf16", "f16x3"])
@pytest.mark.parametrize("ar_mode", ["full", "incremental"])
def test_single_clip_equals_row_of_a_batch(precision, ar_mode):
    """The reference samples ONE clip per call (main_mage.py:205,239-241: DataLoader(batch_size=1)).  B = 1 takes other tile shapes
    than B = 4 (one 128-row tile list vs the 8-phase kernel); the result must be the same clip, bitwise."""
    L = 16
    m = build_mage(synth.mnist_model_config(frames_length=L), 3, DEV).set_precision(precision)
    m.ar_mode = ar_mode
    batch = dev_batch(synth.synth_batch_mnist(4, L, seed=9))
    v4 = m.autoregressive_generate(batch)
    t4 = m.last_tokens.clone()
    for r in (0, 3):
        one = {k: v[r:r + 1] for k, v in batch.items()}
        v1 = m.autoregressive_generate(one)
        assert torch.equal(m.last_tokens, t4[r:r + 1]) and torch.equal(v1, v4[r:r + 1])


# ------------------------------------------------------------------------------------------------ CATER configs at FULL size
def test_full_size_properties_cfg4_caterv1():
    """BASELINE cfg4 (config/mage_caterv1.yaml, frames_length 32, 128x128, B=32, bf16) at its stated size, through
    size-independent properties: bitwise determinism, shard == slice of the whole batch, incremental == full, range."""
    B, L = 32, 32
    m = build_mage(synth.cater_model_config(frames_length=L), 0, DEV).set_precision("bf16")
    batch = dev_batch(synth.synth_batch_cater(B, L, seed=1))
    batch["video_noise"] = torch.randn(B, 64, 16, 16, generator=torch.Generator().manual_seed(5)).to(DEV)
    v1 = m.autoregressive_generate(batch)
    tok1 = m.last_tokens.clone()
    assert tuple(v1.shape) == (B, L, 3, 128, 128) and torch.isfinite(v1).all() and v1.abs().max().item() <= 1.0
    assert torch.equal(v1[:, 0], batch["images"][:, 0])
    assert tok1.min().item() >= 0 and tok1.max().item() < 512 and tok1.unique().numel() > 16       # not a collapsed sequence
    v2 = m.autoregressive_generate(batch)
    assert torch.equal(tok1, m.last_tokens) and torch.equal(v1, v2)
    half = {k: v[B // 2:] for k, v in batch.items()}
    vh = m.autoregressive_generate(half)
    assert torch.equal(m.last_tokens, tok1[B // 2:]) and torch.equal(vh, v1[B // 2:])
    del v2, vh
    m.ar_mode = "incremental"
    vi = m.autoregressive_generate(batch)
    assert torch.equal(m.last_tokens, tok1) and torch.equal(vi, v1)


def test_full_size_properties_cfg5_mage_plus():
    """BASELINE cfg5 (config/mage+_caterv2.yaml, frames_length 32, MAGE+ over a latent first stage, 16 clips = one GPU's share
    of the global 128), MAGE side at its stated size: determinism, shard == slice, finite latents and frames."""
    B, L = 16, 32
    m = build_mage(synth.magep_model_config(frames_length=L), 0, DEV).set_precision("bf16")
    batch = dev_batch(synth.synth_batch_cater(B, L, seed=2, vocab=50, text_len=24))
    batch["video_noise"] = torch.randn(B, 64, 16, 16, generator=torch.Generator().manual_seed(6)).to(DEV)
    v1 = m.autoregressive_generate(batch)
    lat1 = m.last_logits.clone()
    assert tuple(v1.shape) == (B, L, 3, 128, 128) and tuple(lat1.shape) == (B, L - 1, 16, 16, 4)
    assert torch.isfinite(v1).all() and torch.isfinite(lat1).all() and lat1.abs().max().item() > 0
    v2 = m.autoregressive_generate(batch)
    assert torch.equal(v1, v2) and torch.equal(lat1, m.last_logits)
    half = {k: v[:B // 2] for k, v in batch.items()}
    vh = m.autoregressive_generate(half)
    assert torch.equal(vh, v1[:B // 2]) and torch.equal(m.last_logits, lat1[:B // 2])


# ------------------------------------------------------------------------------------------------ HIP-graph replay of the call
@pytest.mark.parametrize("precision,ar_mode", [("fp32", "full"), ("bf16", "full"), ("bf16", "incremental"), ("f16x3", "incremental")])
def test_graph_replay_is_bit_identical_to_eager(precision, ar_mode):
    """MAGE.use_graph: the whole autoregressive_generate call replayed from one captured HIP graph (no allocation, no host
    decision inside) gives bitwise the eager result, also on NEW inputs of the same shape, and the golden tokens in fp32."""
    g = golden("mage_mnist_L6_ragged")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    m = build_mage(synth.mnist_model_config(frames_length=L), seed, DEV).set_precision(precision)
    m.ar_mode = ar_mode
    m.use_graph = False                                     # the eager reference (the default, None, replays by itself at this size)
    kw = dict(digits=int(g["digits"]), text_len=int(g["text_len"]), ragged_text=True)
    b1 = dev_batch(synth.synth_batch_mnist(B, L, seed=seed, **kw))
    b2 = dev_batch(synth.synth_batch_mnist(B, L, seed=seed + 1, **kw))
    want = []
    for b in (b1, b2):
        v = m.autoregressive_generate(b)
        want.append((v.clone(), m.last_tokens.clone()))
    m.use_graph = True
    for rnd_ in range(3):                                   # call 1 eager (warm), call 2 captures + replays, call 3 replays
        for b, (wv, wt) in zip((b1, b2), want):
            v = m.autoregressive_generate(b)
            assert torch.equal(v, wv) and torch.equal(m.last_tokens, wt), (rnd_, m.last_call_mode)
    assert m.last_call_mode == "graph"
    if precision in ("fp32", "f16x3"):
        m.autoregressive_generate(b1)
        assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, "graph-replayed AR tokens") == 0
    # replaced weights invalidate the captured graph (its kernels point at the old derived copies)
    synth.fill_state_dict(m, seed + 7)
    m.use_graph = False
    v_new = m.autoregressive_generate(b1)
    m.use_graph = True
    for _ in range(3):
        v = m.autoregressive_generate(b1)
        assert torch.equal(v, v_new)
    assert not torch.equal(v_new, want[0][0])


def test_graph_replay_is_the_default_for_a_few_clips_per_call():
    """use_graph = None (default): calls of up to 4 clips (the reference samples ONE per call, main_mage.py:205) replay from a captured
    graph from the third call of a shape on -- bitwise the eager result; a changed kernel selection (here: the residual-stream form) or a larger
    batch does not replay a stale graph."""
    m = build_mage(synth.mnist_model_config(frames_length=6), 5, DEV).set_precision("bf16")
    m.ar_mode = "incremental"
    one = dev_batch(synth.synth_batch_mnist(1, 6, seed=5))
    assert m.use_graph is None
    m.use_graph = False
    want = m.autoregressive_generate(one).clone()
    m.use_graph = None
    for _ in range(4):                                      # two eager calls, the third captures + replays, the fourth replays
        v = m.autoregressive_generate(one)
        assert torch.equal(v, want)
    assert m.last_call_mode == "graph"
    m.generate_model.stream_bf16 = False                    # other kernels: a new capture, not the old graph
    m.use_graph = False
    want2 = m.autoregressive_generate(one).clone()
    m.use_graph = None
    for _ in range(4):
        assert torch.equal(m.autoregressive_generate(one), want2)
    assert m.last_call_mode == "graph"
    if config.get().stream_16bit:                           # (MAGE_STREAM_FP32=1 pins both forms to the fp32 stream)
        assert not torch.equal(want, want2)
    big = dev_batch(synth.synth_batch_mnist(8, 6, seed=5))
    for _ in range(3):
        m.autoregressive_generate(big)
    assert m.last_call_mode == "eager"


def test_graph_replay_carries_per_kernel_events():
    """bench.py's roofline needs HIP events around the dominant kernel INSIDE the timed region: under graph replay they are
    event-record nodes of the captured graph, re-recorded by every replay -- where the runtime has them; where it has not (ROCm 7.2),
    profiled calls fall back to eager launches (tested in the first branch)."""
    from mage_amd import ops
    m = build_mage(synth.mnist_model_config(frames_length=4), 5, DEV).set_precision("bf16")
    b = dev_batch(synth.synth_batch_mnist(8, 4, seed=5))
    if not ops.graph_events_supported(DEV):
        # ROCm 7.2: timing events cannot become event-record nodes.  The documented fallback: a call that wants per-launch events runs
        # EAGERLY even with graph replay on (so bench.py's roofline events are always real), and replays again once profiling is off
        try:
            m.use_graph = True
            for _ in range(3):
                m.autoregressive_generate(b)
            assert m.last_call_mode == "graph"
            ref_tok = m.last_tokens.clone()
            ops.PROFILE.reset(enabled=True, only=["layernorm"])
            m.autoregressive_generate(b)
            assert m.last_call_mode == "eager" and ops.PROFILE.summary()["layernorm"]["calls"] > 0 and torch.equal(m.last_tokens, ref_tok)
            ops.PROFILE.reset()
            m.autoregressive_generate(b)
            assert m.last_call_mode == "graph" and torch.equal(m.last_tokens, ref_tok)
        finally:
            ops.PROFILE.reset()
        return
    try:
        ops.PROFILE.reset(enabled=True, only=["layernorm"])
        m.autoregressive_generate(b)           # eager, graph off: the number of bracketed launches per call
        per_call = ops.PROFILE.summary()["layernorm"]["calls"]
        ops.PROFILE.clear()
        m.use_graph = True
        m.autoregressive_generate(b)           # eager (warm)
        m.autoregressive_generate(b)           # capture + replay
        ops.PROFILE.clear()
        m.autoregressive_generate(b)
        m.autoregressive_generate(b)
        assert m.last_call_mode == "graph"
        s = ops.PROFILE.summary()
        assert set(s) == {"layernorm"} and s["layernorm"]["calls"] == 2 * per_call and 0.0 < s["layernorm"]["ms"] < 1e3
    finally:
        ops.PROFILE.reset()


@pytest.mark.parametrize("precision", ["fp32", "bf16", "f16x3"])
def test_frame_table_path_against_the_convolution_path(precision):
    """conv3x3(token embedding) + in_linear as a table sum (MAGE._frame_tables, the default) against the reference's operation order
    (embedding -> convolution GEMM -> in_linear GEMM; model.frame_table = False): same tokens on the golden batch in the parity
    modes, logits within fp32 rounding (fp32 / f16x3) resp. the bf16 error (bf16: the table path is the more exact of the two)."""
    g = golden("mage_mnist_L6_ragged")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    m = build_mage(synth.mnist_model_config(frames_length=L), seed, DEV).set_precision(precision)
    batch = dev_batch(synth.synth_batch_mnist(B, L, seed=seed, digits=int(g["digits"]), text_len=int(g["text_len"]), ragged_text=True))
    assert m._frame_tables()["ft.T2"] is not None
    tok_t, lg_t = m.teacher_forced_logits(batch)
    m.frame_table = False
    tok_c, lg_c = m.teacher_forced_logits(batch)
    assert torch.equal(tok_t, tok_c)
    err = (lg_t - lg_c).abs().max().item()
    print(f"{precision}: teacher-forced logits, table path vs convolution path: {err:.2e}")
    assert err < (2e-5 if precision != "bf16" else 0.06)
    if precision != "bf16":
        for ft in (False, True):
            m.frame_table = ft
            m.autoregressive_generate(batch)
            assert assert_tokens(m.last_tokens.cpu(), g["gen_tokens"], g["margin"], TOK_TOL, f"AR tokens, frame_table={ft}") == 0


def test_f4_encoder_split_precision_path_against_exact_fp32_path():
    """The f4 encoder's convolutions on f16x3 operands (VectorQuantizedVAE._encode_f4_split: every precision except 'fp32', whose
    encoder is the exact chain -- set_precision flips encode_split) against the exact-fp32 MFMA gather path on 64 synthetic frames: features within fp32 rounding of each other, identical tokens wherever
    the quantiser's own top-2 margin exceeds that noise -- and the reference's golden tokens with BOTH (test_vqvae_f4_golden...)."""
    m = build_vqvae(1, 4, 256, 512, 17, DEV)
    x = synth.synth_batch_mnist(4, 16, seed=23)["images"].reshape(64, 1, 64, 64).to(DEV)
    assert m.encode_split is False                     # a fresh model is in 'fp32' mode: the exact encoder
    m.set_precision("bf16")
    assert m.encode_split is True
    z_s = m._encode_features(x).clone()
    ids_s = m.encode(x)
    m.set_precision("fp32")
    assert m.encode_split is False
    z_f = m._encode_features(x)
    ids_f = m.encode(x)
    err = (z_s - z_f).abs().max().item()
    w = m._weights()
    from mage_amd import ops as o
    _, margin = o.vq_nearest(z_f, w["cbt"], w["c2"], want_margin=True)
    bad = (ids_s.reshape(-1) != ids_f.reshape(-1))
    print(f"f16x3 encoder vs exact fp32: max |d z_e| {err:.2e} (|z_e| max {z_f.abs().max().item():.2f}), token mismatches {int(bad.sum())} of {bad.numel()}")
    assert err < 3e-5
    assert not (bad & (margin > TOK_TOL)).any()


def test_tokens_on_trained_weights_follow_the_oracle_outside_the_error_margin():
    """north_star: reference-matching token sequences.  On random-init weights the decoder's top-2 margins (1e-5 class) are below any 16-bit
    mode's logit error, so those modes cannot match there; this test gives a small model a trained model's margins -- stage 1 (VQ-VAE) and
    stage 2 (MAGE) trained IN-TREE on the HIP training path (bench.py's trained_token_agreement leg at a reduced size, the 'strokes' task) --
    and compares held-out clips with the CPU oracle on the same weights:
      f16x3: identical tokens;
      f16 / bf16, teacher-forced on the oracle's sequence (per-decision, no compounding): every disagreement sits at a decision the oracle
        took by less than twice the mode's measured logit error; f16's error is a fraction of bf16's and its per-decision agreement >= 0.999;
      free-running: a clip leaves the oracle's sequence first at such a decision (every later position of that clip sees other inputs);
        f16 keeps at least as many positions as bf16."""
    import bench
    r = bench.trained_token_agreement(torch.device(DEV), 6, 200, 16, 4, 8, cfg_kw=dict(width=128, layers=3, vq_dim=64, K=64))
    print({k: v for k, v in r.items() if k != "note"})
    assert r["stage1"]["distinct_tokens_in_the_oracle_sequences"] >= 8          # a non-degenerate token task
    assert r["loss_trajectory"][-1] < 0.5 * r["loss_trajectory"][0]
    assert r["f16x3"]["all_positions"] == 1.0
    for prec in ("bf16", "f16"):
        assert r[prec]["first_divergences_inside_twice_the_modes_logit_error"] is True
        assert r[prec]["teacher_forced"]["mismatches_where_oracle_margin_above_twice_the_error"] == 0
    assert r["f16"]["teacher_forced_max_logit_error_vs_oracle"] < 0.3 * r["bf16"]["teacher_forced_max_logit_error_vs_oracle"]
    assert r["f16"]["teacher_forced"]["token_agreement"] >= 0.999
    assert r["f16"]["all_positions"] >= r["bf16"]["all_positions"] - 1e-9
    # reproducible: seeded dropout, fixed-order gradient sums (mage_embedding_bwd's deterministic form) -> the same trained weights again
    r2 = bench.trained_token_agreement(torch.device(DEV), 6, 200, 16, 4, 8, cfg_kw=dict(width=128, layers=3, vq_dim=64, K=64))
    assert r2["trained_weights_sha256"] == r["trained_weights_sha256"]
    assert r2["f16"]["all_positions"] == r["f16"]["all_positions"] and r2["bf16"]["all_positions"] == r["bf16"]["all_positions"]
