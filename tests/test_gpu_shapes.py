"""Shape generality of the HIP path: the reference takes `image_resolution`, `vision_width`, the batch size and the
caption length as free arguments (/root/reference/modules/mage_model.py:447-514); every other GPU test runs image_resolution = 16 and
vision_width in {64, 128, 512}.  Here the same end-to-end comparison with the CPU oracle (fp32: tokens exact outside fp32 rounding noise,
logits / frames / loss within 1e-4) at shapes that fall OFF the fused fast paths -- frames that are not whole 256-row tiles, widths that are
not multiples of 256, 32 x 32 token grids, captions that fill the context -- and, per shape, the 16-bit and split modes: they must run (a
fused path either takes the shape or hands it to the generic kernel) and the incremental loop must stay bit-identical to the full loop."""
import numpy as np
import pytest
import torch

from mage_amd.utils import synth
from oracle import mage_oracle as O
from tests.helpers import assert_tokens, build_mage, cpu_sd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOK_TOL = 2e-5
LOGIT_TOL = 1e-4

#        name             R   width layers B  L  caption
CASES = [("R8_B3_w512",    8,  512,  6,    3, 4, 11),      # 192 rows per frame slot: no whole 256-row tile anywhere
         ("R8_B1_w512",    8,  512,  3,    1, 5, 11),      # one clip: 64 rows per slot (the few-rows kernels)
         ("R32_w256",      32, 256,  3,    1, 3, 11),      # 128 x 128 inputs on the f4 VQ-VAE, 1024 tokens per frame
         ("R16_w256",      16, 256,  3,    2, 4, 11),
         ("R16_w768",      16, 768,  3,    2, 4, 11),      # 24 heads, not a multiple of 256 or 512 columns
         ("R16_w320",      16, 320,  3,    2, 4, 11),      # 10 heads, a multiple of 64 only
         ("R16_caption32", 16, 512,  3,    2, 4, 32),      # captions as long as the context (a9's key-padding path with no padding at all)
         ("R24_w512",      24, 512,  3,    1, 3, 13)]      # 96 x 96 inputs: 576 tokens per frame (2.25 tiles)


def make_batch(B, L, R, S, seed, vocab=30):
    """Synthetic clip batch at 4R x 4R pixels with the reference's batch contract (dataload.py:240-271): smooth random blobs moving by a few pixels
    per frame, pixel range [-0.5, 0.5]; captions [CLS] .. [SEP] of full length S for the first clip, right-padded for the others."""
    g = torch.Generator().manual_seed(seed)
    px = 4 * R
    base = torch.rand(B, 1, px // 4 + 8, px // 4 + 8, generator=g)
    base = torch.nn.functional.interpolate(base, scale_factor=4, mode="bilinear", align_corners=False)
    imgs = torch.empty(B, L, 1, px, px)
    for t_ in range(L):
        o = 2 * t_
        imgs[:, t_] = base[:, :, o:o + px, o:o + px]
    imgs = (imgs > 0.55).float() * imgs - 0.5
    text = torch.zeros(B, S, dtype=torch.int64)
    for b in range(B):
        n = S if b == 0 else max(4, S - 2 * b)
        text[b, 0] = 1
        text[b, 1:n - 1] = torch.randint(3, vocab, (n - 2,), generator=g)
        text[b, n - 1] = 2
    return {"images": imgs.contiguous(), "text": text, "speed": torch.rand(B, generator=g)}


@pytest.mark.parametrize("name,R,width,layers,B,L,S", CASES, ids=[c[0] for c in CASES])
def test_shape_against_the_oracle_and_across_modes(name, R, width, layers, B, L, S):
    cfg = synth.mnist_model_config(frames_length=L, width=width, layers=layers, image_resolution=R, context_length=32)
    m = build_mage(cfg, 11, DEV)
    batch = make_batch(B, L, R, S, seed=100 + R + width)
    db = {k: v.to(DEV) for k, v in batch.items()}
    sd = cpu_sd(m)
    with torch.no_grad():
        o_video, o_tok, o_tok0, o_trace = O.mage_generate(sd, batch, L, return_trace=True)
        o_loss, _ = O.mage_forward_loss(sd, batch, L)
    top2 = o_trace.topk(2, dim=-1)[0]
    margin = (top2[..., 0] - top2[..., 1]).numpy()
    # ---- fp32: the parity mode
    video = m.autoregressive_generate(db)
    assert tuple(video.shape) == (B, L, 1, 4 * R, 4 * R)
    n_soft = assert_tokens(m.last_tokens.cpu(), o_tok, margin, TOK_TOL, f"{name} fp32 AR tokens")
    if n_soft == 0:
        torch.testing.assert_close(video.cpu(), o_video, atol=LOGIT_TOL, rtol=0)
        torch.testing.assert_close(m.last_logits.cpu(), o_trace, atol=LOGIT_TOL, rtol=0)      # per-step logits == the last pass's, by causality
    loss, _ = m(db)
    assert abs(loss.item() - o_loss.item()) < 1e-4, (loss.item(), o_loss.item())
    t_full32 = m.last_tokens.clone() if m.last_tokens is not None else None
    m.ar_mode = "incremental"
    v_inc = m.autoregressive_generate(db)
    assert torch.equal(v_inc, video), f"{name}: fp32 incremental loop differs from the full loop"
    # ---- the other modes: run, incremental == full bitwise, f16x3 follows the oracle like fp32 does
    for prec in ("f16x3", "bf16", "f16"):
        m.set_precision(prec)
        m.ar_mode = "full"
        v_full = m.autoregressive_generate(db)
        t_full = m.last_tokens.clone()
        m.ar_mode = "incremental"
        v_i = m.autoregressive_generate(db)
        assert torch.equal(m.last_tokens, t_full) and torch.equal(v_i, v_full), f"{name} {prec}: incremental loop differs from the full loop"
        if prec == "f16x3":
            assert_tokens(t_full.cpu(), o_tok, margin, 1e-4, f"{name} f16x3 AR tokens")
        else:
            assert torch.isfinite(v_full).all()
            agree = (t_full.cpu() == o_tok).float().mean().item()
            print(f"{name} {prec}: free-running token agreement with the oracle {agree:.4f} (random-init margins: informational)")
    del t_full32


def test_vqvae_f4_at_other_resolutions_against_the_oracle():
    """The f4 first stage alone at 32 x 32, 96 x 96 and 128 x 128 pixels and odd frame counts: ids bit-exact outside the near-tie margin, frames 1e-4."""
    from tests.helpers import build_vqvae
    m = build_vqvae(1, 4, 256, 512, 3, DEV)
    sd = {"first_stage_model." + k: v for k, v in cpu_sd(m).items()}
    for n, px in ((5, 32), (3, 96), (1, 128), (7, 48)):
        g = torch.Generator().manual_seed(px)
        x = torch.rand(n, 1, px, px, generator=g) - 0.5
        with torch.no_grad():
            z = O.vqvae_encoder(sd, "first_stage_model.", x).permute(0, 2, 3, 1)
            d = O.vq_distances(z.reshape(-1, z.shape[-1]), sd["first_stage_model.codebook.embedding.weight"])
            ids_o = d.argmin(1).view(n, px // 4, px // 4)
            t2 = d.topk(2, dim=1, largest=False)[0]
            mg = (t2[:, 1] - t2[:, 0]).numpy()
            rec_o = O.vqvae_decode(sd, "first_stage_model.", ids_o)
        ids = m.encode(x.to(DEV))
        assert tuple(ids.shape) == (n, px // 4, px // 4)
        assert_tokens(ids.cpu(), ids_o, mg, TOK_TOL, f"f4 encode {px}px")
        rec = m.decode(ids_o.to(DEV))
        torch.testing.assert_close(rec.cpu(), rec_o, atol=LOGIT_TOL, rtol=0)


@pytest.mark.parametrize("R,width,B,L", [(8, 256, 3, 3), (24, 512, 1, 3)], ids=["R8_w256_B3", "R24_w512"])
def test_cater_f8_randomness_branch_at_other_resolutions(R, width, B, L):
    """cfg4's model family (f8 VQ-VAE on RGB, ADAIN sampling branch, mage_model.py:299-314,660-664) at 64 x 64 and 192 x 192 pixels."""
    cfg = synth.cater_model_config(frames_length=L, width=width, layers=3, vq_dim=64, K=128)
    cfg["params"]["image_resolution"] = R
    m = build_mage(cfg, 5, DEV)
    g = torch.Generator().manual_seed(R)
    px = 8 * R
    small = torch.rand(B, L, 3, px // 8, px // 8, generator=g) * 2 - 1
    imgs = torch.nn.functional.interpolate(small.view(B * L, 3, px // 8, px // 8), scale_factor=8, mode="nearest").view(B, L, 3, px, px).contiguous()
    text = torch.zeros(B, 12, dtype=torch.int64)
    text[:, 0], text[:, 1:11], text[:, 11] = 1, torch.randint(3, 30, (B, 10), generator=g), 2
    batch = {"images": imgs, "text": text, "speed": torch.rand(B, generator=g)}
    noise = torch.randn(B, 64, R, R, generator=g)
    with torch.no_grad():
        o_video, o_tok, _, o_trace = O.mage_generate(cpu_sd(m), batch, L, noise=noise, return_trace=True)
    top2 = o_trace.topk(2, dim=-1)[0]
    margin = (top2[..., 0] - top2[..., 1]).numpy()
    db = {k: v.to(DEV) for k, v in batch.items()}
    db["video_noise"] = noise.to(DEV)
    video = m.autoregressive_generate(db)
    if assert_tokens(m.last_tokens.cpu(), o_tok, margin, TOK_TOL, f"cater R={R} AR tokens") == 0:
        torch.testing.assert_close(m.last_logits.cpu(), o_trace, atol=LOGIT_TOL, rtol=0)
        torch.testing.assert_close(video.cpu(), o_video, atol=LOGIT_TOL, rtol=0)
    for prec in ("f16x3", "bf16", "f16"):
        m.set_precision(prec)
        m.ar_mode = "full"
        v_full = m.autoregressive_generate(db)
        t_full = m.last_tokens.clone()
        m.ar_mode = "incremental"
        v_i = m.autoregressive_generate(db)
        assert torch.equal(m.last_tokens, t_full) and torch.equal(v_i, v_full), f"cater R={R} {prec}: incremental loop differs from the full loop"
