"""CPU: pin the oracle (oracle/mage_oracle.py) against golden vectors produced by the reference itself
(tools/gen_golden.py).  No GPU, no /root/reference needed."""
import numpy as np
import pytest
import torch

from mage_amd.utils import synth
from oracle import mage_oracle as O
from tests.helpers import build_mage, build_vqvae, chk, cpu_sd, golden, t

torch.set_num_threads(8)


def test_vq_unit_ties_first_index():
    g = golden("vq_unit")
    idx = O.vq_nearest(t(g["z"]), t(g["cb"]))
    assert torch.equal(idx, t(g["idx"]))
    assert idx[0].item() == 2 and idx[1].item() == 9          # duplicated codes: the lower index wins
    idx2 = O.vq_nearest(t(g["z2"]), t(g["cb2"]))
    assert torch.equal(idx2, t(g["idx2"]))


def test_vqvae_f4_encode_decode():
    g = golden("vqvae_f4")
    sd = {"vq." + k: v for k, v in cpu_sd(build_vqvae(1, 4, 256, 512, int(g["seed"]))).items()}
    x = t(g["x"])
    z = O.vqvae_encoder(sd, "vq.", x)
    assert torch.allclose(z[:, :8], t(g["z_e_slice"]), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(chk(z), g["z_e_chk"], rtol=1e-5)
    ids = O.vqvae_encode(sd, "vq.", x)
    assert torch.equal(ids, t(g["ids"]).long())
    rec = O.vqvae_decode(sd, "vq.", ids)
    assert torch.allclose(rec, t(g["rec"]), atol=1e-5)


def test_vqvae_f8_encode_decode():
    g = golden("vqvae_f8")
    sd = {"vq." + k: v for k, v in cpu_sd(build_vqvae(3, 8, int(g["dim"]), int(g["K"]), int(g["seed"]))).items()}
    x = synth.synth_batch_cater(2, 1, seed=int(g["seed"]))["images"][:, 0].contiguous()
    z = O.vqvae_encoder(sd, "vq.", x)
    assert torch.allclose(z[:, :8], t(g["z_e_slice"]), atol=2e-5, rtol=1e-5)
    ids = O.vqvae_encode(sd, "vq.", x)
    assert torch.equal(ids, t(g["ids"]).long())
    rec = O.vqvae_decode(sd, "vq.", ids)
    assert torch.allclose(rec[..., ::4, ::4], t(g["rec_sub"]), atol=1e-5)
    np.testing.assert_allclose(chk(rec), g["rec_chk"], rtol=1e-5)


@pytest.mark.parametrize("tag", ["mage_mnist_L4", "mage_mnist_L6_ragged"])
def test_mage_full_width_stages(tag):
    g = golden(tag)
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    sd = cpu_sd(build_mage(synth.mnist_model_config(frames_length=L), seed))
    batch = synth.synth_batch_mnist(B, L, seed=seed, digits=int(g["digits"]), text_len=int(g["text_len"]),
                                    ragged_text=bool(g["ragged"]))
    assert torch.equal(batch["text"], t(g["text"]))
    txt = O.text_encoder(sd, "text_encoder.", batch["text"])
    assert torch.allclose(txt, t(g["text_emb"]), atol=2e-5, rtol=1e-5)
    tok0 = O.vqvae_encode(sd, "first_stage_model.", batch["images"][:, 0])
    assert torch.equal(tok0, t(g["tok0"]).long())
    ma = O.motion_anchor(sd, tok0, batch["text"], batch["speed"])
    assert torch.allclose(ma[:, ::4, ::4], t(g["motion_sub"]), atol=5e-5, rtol=1e-5)
    video, gen, _, trace = O.mage_generate(sd, batch, L, return_trace=True)
    assert torch.allclose(trace[:, :, ::4, ::4], t(g["step_logits_sub"]), atol=1e-4, rtol=1e-4)
    assert torch.equal(gen, t(g["gen_tokens"]).long())
    assert torch.allclose(video, t(g["video"]), atol=1e-5)
    loss, _ = O.mage_forward_loss(sd, batch, L)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    assert set(g["loss_dict_keys"].tolist()) == {"val/prediction", "val/final_loss"}


def test_mage_reduced_width_d64():
    g = golden("mage_small_d64")
    cfg = synth.mnist_model_config(frames_length=int(g["L"]), width=64, layers=3, vq_dim=32, K=64)
    sd = cpu_sd(build_mage(cfg, int(g["seed"])))
    batch = synth.synth_batch_mnist(int(g["B"]), int(g["L"]), seed=int(g["seed"]), text_len=int(g["text_len"]), ragged_text=True)
    video, gen, _, trace = O.mage_generate(sd, batch, int(g["L"]), return_trace=True)
    assert torch.allclose(trace, t(g["step_logits"]), atol=1e-4, rtol=1e-4)
    assert torch.equal(gen, t(g["gen_tokens"]).long())
    assert torch.allclose(video, t(g["video"]), atol=1e-5)


def test_mage_cater_small_randomness_branch():
    g = golden("mage_cater_small")
    cfg = synth.cater_model_config(frames_length=int(g["L"]), width=64, layers=3, vq_dim=32, K=64)
    sd = cpu_sd(build_mage(cfg, int(g["seed"])))
    batch = synth.synth_batch_cater(int(g["B"]), int(g["L"]), seed=int(g["seed"]), text_len=int(g["text_len"]))
    video, gen, tok0, trace = O.mage_generate(sd, batch, int(g["L"]), noise=t(g["noise"]), return_trace=True)
    ma = O.motion_anchor(sd, tok0, batch["text"], batch["speed"], t(g["noise"]))
    assert torch.allclose(ma, t(g["motion"]), atol=1e-4, rtol=1e-4)
    assert torch.allclose(trace, t(g["step_logits"]), atol=1e-5, rtol=1e-5)      # measured: 1.8e-6
    assert torch.equal(gen, t(g["gen_tokens"]).long())
    assert torch.allclose(video[..., ::4, ::4], t(g["video_sub"]), atol=1e-5)


def test_mage_cater_forward_randomness_losses():
    """MAGE.forward with randomness=True (Conv3d video prior, reparameterisation with the reference's noise injected, KL and
    speed-l2 terms) against the reference's own (loss, loss_dict), its conv3d output and its teacher-forced logits."""
    g = golden("mage_cater_forward_small")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    cfg = synth.cater_model_config(frames_length=L, width=int(g["width"]), layers=int(g["layers"]), vq_dim=int(g["vq_dim"]), K=int(g["K"]))
    sd = cpu_sd(build_mage(cfg, seed))
    batch = synth.synth_batch_cater(B, L, seed=seed, text_len=int(g["text_len"]))
    final, parts, logits, prior = O.mage_forward_loss_random(sd, batch, L, t(g["eps"]), alpha=cfg["params"]["alpha"],
                                                             beta=cfg["params"]["beta"])
    assert torch.allclose(prior[:, ::4, ::2, ::2], t(g["prior_sub"]), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(chk(prior), g["prior_chk"], rtol=1e-5)
    assert torch.allclose(logits[:, ::3, ::4, ::4, ::8], t(g["logits_sub"]), atol=1e-4, rtol=1e-4)
    assert abs(parts["prediction"] - float(g["prediction"])) < 1e-5
    assert abs(parts["kl_loss"] - float(g["kl_loss"])) < 1e-4 * max(1.0, abs(float(g["kl_loss"])))
    assert abs(final.item() - float(g["final_loss"])) < 1e-5 * max(1.0, abs(float(g["final_loss"])))


def test_mage_plus_forward_latent_losses():
    """MAGE.forward for use_cids=False (MSE on latents, randomness + PID-controlled beta: config/mage+_*.yaml) over the stand-in
    latent first stage, against the reference's own (loss, loss_dict) and predicted latents."""
    from modules.mage_model import PIDControl
    from tests.standin_first_stage import StandInLatentFirstStage
    g = golden("mage_plus_forward_small")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    cfg = synth.magep_model_config(frames_length=L, width=int(g["width"]), layers=int(g["layers"]))
    sd = cpu_sd(build_mage(cfg, seed))
    batch = synth.synth_batch_cater(B, L, seed=seed, text_len=int(g["text_len"]), vocab=50)
    fs = StandInLatentFirstStage()
    lat = fs.encode(batch["images"].reshape(B * L, *batch["images"].shape[2:])).view(B, L, 4, 16, 16)
    final, parts, pred = O.mage_forward_loss_latent(sd, batch, L, lat, t(g["eps"]), v_kl=cfg["params"]["v_kl"], pid=PIDControl())
    assert torch.allclose(pred, t(g["pred"]), atol=1e-4, rtol=1e-4)
    assert abs(parts["prediction"] - float(g["prediction"])) < 1e-5 * max(1.0, abs(float(g["prediction"])))
    assert abs(parts["kl_loss"] - float(g["kl_loss"])) < 1e-4 * max(1.0, abs(float(g["kl_loss"])))
    assert abs(parts["beta"] - float(g["beta"])) < 1e-9
    assert abs(final.item() - float(g["final_loss"])) < 1e-5 * max(1.0, abs(float(g["final_loss"])))


def test_mage_L16_tokens():
    g = golden("mage_mnist_L16")
    sd = cpu_sd(build_mage(synth.mnist_model_config(frames_length=16), int(g["seed"])))
    batch = synth.synth_batch_mnist(int(g["B"]), 16, seed=int(g["seed"]))
    video, gen, _, trace = O.mage_generate(sd, batch, 16, return_trace=True)
    assert torch.equal(gen, t(g["gen_tokens"]).long())
    assert torch.allclose(trace[:, :, ::8, ::8, ::4], t(g["step_logits_sub"]), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(chk(video), g["video_chk"], rtol=1e-5)


def test_mage_plus_latent_path():
    """MAGE+ side (use_cids=False): Linear(4->C) in, GroupNorm/SiLU/Conv3d head out, over the stand-in latent first stage."""
    from tests.standin_first_stage import StandInLatentFirstStage
    g = golden("mage_plus_small")
    cfg = synth.magep_model_config(frames_length=int(g["L"]), width=64, layers=3)
    sd = cpu_sd(build_mage(cfg, int(g["seed"])))
    batch = synth.synth_batch_cater(int(g["B"]), int(g["L"]), seed=int(g["seed"]), text_len=int(g["text_len"]), vocab=50)
    fs = StandInLatentFirstStage()
    lat0 = fs.encode(batch["images"][:, 0])
    pred = O.mage_generate_latent(sd, batch, int(g["L"]), lat0, noise=t(g["noise"]))
    assert torch.allclose(pred, t(g["pred_latents"]), atol=1e-4, rtol=1e-4)
    video = fs.decode(pred.reshape(-1, 16, 16, 4).permute(0, 3, 1, 2)).view(int(g["B"]), -1, 3, 128, 128)
    assert torch.allclose(video[..., ::4, ::4], t(g["video_sub"])[:, 1:], atol=1e-4)


def test_mage_cater_fullwidth_golden():
    """cfg4's model at FULL width (d=512, 6 blocks, f8 VQ-VAE dim 256 -> codebook D=1024, K=512, randomness branch) on a short
    clip: the oracle against the reference's own tokens, logits, motion anchor and frames."""
    g = golden("mage_cater_fullwidth")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    sd = cpu_sd(build_mage(synth.cater_model_config(frames_length=L), seed))
    batch = synth.synth_batch_cater(B, L, seed=seed, text_len=int(g["text_len"]))
    video, gen, tok0, trace = O.mage_generate(sd, batch, L, noise=t(g["noise"]), return_trace=True)
    assert torch.equal(tok0, t(g["tok0"]).long()) and torch.equal(gen, t(g["gen_tokens"]).long())
    ma = O.motion_anchor(sd, tok0, batch["text"], batch["speed"], t(g["noise"]))
    assert torch.allclose(ma[:, ::4, ::4], t(g["motion_sub"]), atol=2e-5, rtol=1e-5)
    assert torch.allclose(trace[:, :, ::4, ::4], t(g["step_logits_sub"]), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(chk(trace), g["step_logits_chk"], rtol=1e-5)
    assert torch.allclose(video[..., ::4, ::4], t(g["video_sub"]), atol=1e-5)
    np.testing.assert_allclose(chk(video), g["video_chk"], rtol=1e-5)


def test_mage_plus_transformer_block_variant():
    """The MAGE+ variant of TransformerBlock.forward (mage_model.py:93: ln_q / ln_kv applied; the fixture comes from the
    reference with that documented line swapped in, in the generating process only): sampling and the teacher-forced loss."""
    from modules.mage_model import PIDControl
    from tests.standin_first_stage import StandInLatentFirstStage
    g = golden("mage_plus_block_small")
    B, L = int(g["B"]), int(g["L"])
    cfg = synth.magep_model_config(frames_length=L, width=int(g["width"]), layers=int(g["layers"]))
    sd = cpu_sd(build_mage(cfg, int(g["seed"])))
    batch = synth.synth_batch_cater(B, L, seed=int(g["seed"]), text_len=int(g["text_len"]), vocab=50)
    fs = StandInLatentFirstStage()
    pred, ma = O.mage_generate_latent(sd, batch, L, fs.encode(batch["images"][:, 0]), noise=t(g["noise"]), mage_plus=True,
                                      return_motion=True)
    assert torch.allclose(ma, t(g["motion"]), atol=2e-5, rtol=1e-5)
    assert torch.allclose(pred, t(g["pred_latents"]), atol=2e-5, rtol=1e-5)
    # without the variant the same weights give a visibly different anchor: the fixture does distinguish the two lines
    _, ma92 = O.mage_generate_latent(sd, batch, L, fs.encode(batch["images"][:, 0]), noise=t(g["noise"]), mage_plus=False,
                                     return_motion=True)
    assert (ma92 - t(g["motion"])).abs().max().item() > 1e-2
    Lf, fseed = int(g["fwd_L"]), int(g["fwd_seed"])
    cfgf = synth.magep_model_config(frames_length=Lf, width=int(g["width"]), layers=int(g["layers"]))
    sdf = cpu_sd(build_mage(cfgf, fseed))
    bf = synth.synth_batch_cater(B, Lf, seed=fseed, text_len=int(g["text_len"]), vocab=50)
    lat = fs.encode(bf["images"].reshape(B * Lf, *bf["images"].shape[2:])).view(B, Lf, 4, 16, 16)
    final, parts, predf = O.mage_forward_loss_latent(sdf, bf, Lf, lat, t(g["fwd_eps"]), v_kl=cfgf["params"]["v_kl"], pid=PIDControl(),
                                                     mage_plus=True)
    assert torch.allclose(predf[:, ::3], t(g["fwd_pred_sub"]), atol=2e-5, rtol=1e-5)
    assert abs(parts["prediction"] - float(g["fwd_prediction"])) < 1e-5 * max(1.0, abs(float(g["fwd_prediction"])))
    assert abs(parts["kl_loss"] - float(g["fwd_kl_loss"])) < 1e-4 * max(1.0, abs(float(g["fwd_kl_loss"])))
    assert abs(parts["beta"] - float(g["fwd_beta"])) < 1e-9
    assert abs(final.item() - float(g["fwd_final_loss"])) < 1e-5 * max(1.0, abs(float(g["fwd_final_loss"])))


def _vq_train_sd(dim, K, seed):
    from mage_amd.modules.vqvae_model import VectorQuantizedVAE
    m = VectorQuantizedVAE(1, 4, dim, K)
    synth.fill_state_dict(m, seed)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def check_vq_train_grads(g, grads, tol):
    """Every parameter gradient of the reference's training step: in full where the fixture holds it, else slices + checksums."""
    worst = 0.0
    names = g["param_names"].tolist()
    gmax_all = max(float(g["gmax." + n]) for n in names)
    for n in names:
        got = grads[n].detach().float().cpu()
        scale = float(g["gmax." + n])
        if scale < 1e-4 * gmax_all:
            # a bias in front of a training-mode BatchNorm has an analytically ZERO gradient (the norm removes the mean): the
            # reference's own values there are rounding noise (1e-10 .. 1e-6); ours must be noise too
            assert got.abs().max().item() < 1e-4 * gmax_all, n
            continue
        if ("g." + n) in g.files:
            err = (got - t(g["g." + n])).abs().max().item() / scale
        else:
            want = t(g["gs." + n])
            err = (got.flatten()[::max(1, got.numel() // 1024)][:1024] - want).abs().max().item() / scale
            np.testing.assert_allclose(chk(got)[1:], g["gchk." + n][1:], rtol=max(tol * 20, 1e-4), atol=1e-5)
        worst = max(worst, err)
        assert err < tol, (n, err)
    return worst


@pytest.mark.parametrize("tag", ["vqvae_f4_train_small", "vqvae_f4_train"])
def test_vqvae_stage1_training_step(tag):
    """The oracle's training-mode VQ-VAE (BatchNorm on batch statistics, straight-through quantiser, in-place-ReLU ResBlocks) and
    its autograd against the reference's own train_vqvae.py step: the three loss terms and every parameter gradient."""
    g = golden(tag)
    sd = {k: (v.requires_grad_() if v.is_floating_point() and "running" not in k else v) for k, v in _vq_train_sd(int(g["dim"]), int(g["K"]),
                                                                                                                   int(g["seed"])).items()}
    x = synth.synth_batch_mnist(int(g["n_img"]), 1, seed=int(g["seed"]))["images"][:, 0].contiguous()
    loss, (rec, vql, com), (x_tilde, z_e, z_q) = O.vqvae_train_loss(sd, "", x, beta=float(g["beta"]))
    assert abs(loss.item() - float(g["loss"])) < 1e-6 and abs(rec.item() - float(g["rec"])) < 1e-6
    assert abs(vql.item() - float(g["vq"])) < 1e-6 * max(1, float(g["vq"])) and abs(com.item() - float(g["commit"])) < 1e-6 * max(1, float(g["commit"]))
    assert torch.allclose(x_tilde[:, :, ::4, ::4], t(g["x_tilde_sub"]), atol=1e-5)
    names = g["param_names"].tolist()
    gs = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
    grads = {n: (gg if gg is not None else torch.zeros_like(sd[n])) for n, gg in zip(names, gs)}
    assert check_vq_train_grads(g, grads, 2e-5) < 2e-5


@pytest.mark.parametrize("tag", ["vqvae_f8_train_small", "vqvae_f8_train"])
def test_vqvae8_stage1_training_step(tag):
    """The oracle's f8 (CATER) VQ-VAE forward with the straight-through quantiser and its autograd against the reference's own
    train_vqvae.py step (--dataset cater-gen: down_ratio 8): the three loss terms and every parameter gradient."""
    from tests.helpers import build_vqvae
    g = golden(tag)
    sd = {k: (v.requires_grad_() if v.is_floating_point() else v) for k, v in cpu_sd(build_vqvae(3, 8, int(g["dim"]), int(g["K"]), int(g["seed"]))).items()}
    x = synth.synth_batch_cater(int(g["n_img"]), 1, seed=int(g["seed"]), res=64)["images"][:, 0].contiguous()
    loss, (rec, vql, com), (x_tilde, z_e, z_q) = O.vqvae_train_loss(sd, "", x, beta=float(g["beta"]))
    assert abs(loss.item() - float(g["loss"])) < 1e-6 * max(1, float(g["loss"])) and abs(rec.item() - float(g["rec"])) < 1e-6
    assert abs(vql.item() - float(g["vq"])) < 1e-6 * max(1, float(g["vq"])) and abs(com.item() - float(g["commit"])) < 1e-6 * max(1, float(g["commit"]))
    assert torch.allclose(x_tilde[:, :, ::4, ::4], t(g["x_tilde_sub"]), atol=1e-5)
    names = g["param_names"].tolist()
    gs = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
    grads = {n: (gg if gg is not None else torch.zeros_like(sd[n])) for n, gg in zip(names, gs)}
    assert check_vq_train_grads(g, grads, 2e-5) < 2e-5
