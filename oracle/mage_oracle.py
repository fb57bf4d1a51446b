"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A from-scratch, functional (no nn.Module) PyTorch-CPU fp32 restatement of the
reference's autoregressive video-token generation path.  Each function cites
the reference file:line whose arithmetic it restates.  It operates directly on
a flat ``state_dict`` (the reference's key layout).

Who may import this: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the timed CPU baseline,
never as the thing shipped.  Nothing under ``mage_amd/`` imports it; the product
path raises if the HIP library is missing instead of falling back here.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md 4), and its arithmetic lives in PyTorch.  This oracle is pinned
against outputs of the reference itself, run in the build container by
``tools/gen_golden.py`` (reference imported read-only) and committed as
fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every
stage.  Exception: the MAGE+ first stage (``ldm`` AutoencoderKL, absent from the
reference mount, unpinned in requirements.txt:21) is NOT covered: parity
unpinned for that component.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- VQ-VAE
def _bn_eval(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """BatchNorm2d in eval mode (first stage is frozen + eval: mage_model.py:516-521)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def resblock(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """vqvae_model.py:111-124.  The leading ReLU is in-place, so the skip path carries
    relu(x): result = relu(x) + BN(conv1x1(relu(BN(conv3x3(relu(x))))))."""
    r = torch.relu(x)
    h = F.conv2d(r, sd[p + ".block.1.weight"], sd[p + ".block.1.bias"], padding=1)
    h = torch.relu(_bn_eval(sd, p + ".block.2", h))
    h = F.conv2d(h, sd[p + ".block.4.weight"], sd[p + ".block.4.bias"])
    return r + _bn_eval(sd, p + ".block.5", h)


def encoder_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """vqvae_model.py:126-145 (non-in-place ReLUs, textbook residual)."""
    idp = F.conv2d(x, sd[p + ".id_path.weight"], sd[p + ".id_path.bias"]) if (p + ".id_path.weight") in sd else x
    h = F.conv2d(torch.relu(x), sd[p + ".block.1.weight"], sd[p + ".block.1.bias"], padding=1)
    h = F.conv2d(torch.relu(h), sd[p + ".block.3.weight"], sd[p + ".block.3.bias"], padding=1)
    h = F.conv2d(torch.relu(h), sd[p + ".block.5.weight"], sd[p + ".block.5.bias"], padding=1)
    h = F.conv2d(torch.relu(h), sd[p + ".block.7.weight"], sd[p + ".block.7.bias"])
    return idp + h


def decoder_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """vqvae_model.py:147-166."""
    idp = F.conv2d(x, sd[p + ".id_path.weight"], sd[p + ".id_path.bias"]) if (p + ".id_path.weight") in sd else x
    h = F.conv2d(torch.relu(x), sd[p + ".block.1.weight"], sd[p + ".block.1.bias"])
    h = F.conv2d(torch.relu(h), sd[p + ".block.3.weight"], sd[p + ".block.3.bias"], padding=1)
    h = F.conv2d(torch.relu(h), sd[p + ".block.5.weight"], sd[p + ".block.5.bias"], padding=1)
    h = F.conv2d(torch.relu(h), sd[p + ".block.7.weight"], sd[p + ".block.7.bias"], padding=1)
    return idp + h


def vqvae_down_ratio(sd: SD, p: str) -> int:
    """f4 nets have a 4x4 stem (vqvae_model.py:172), f8 nets a 7x7 stem (:192)."""
    return 4 if sd[p + "encoder.0.weight"].shape[-1] == 4 else 8


def vqvae_encoder(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """z_e_x [N, D, h, w].  f4: vqvae_model.py:172-179; f8: :192-202."""
    e = p + "encoder"
    if vqvae_down_ratio(sd, p) == 4:
        h = F.conv2d(x, sd[e + ".0.weight"], sd[e + ".0.bias"], stride=2, padding=1)
        h = torch.relu(_bn_eval(sd, e + ".1", h))
        h = F.conv2d(h, sd[e + ".3.weight"], sd[e + ".3.bias"], stride=2, padding=1)
        return resblock(sd, e + ".5", resblock(sd, e + ".4", h))
    h = F.conv2d(x, sd[e + ".0.weight"], sd[e + ".0.bias"], padding=3)
    h = F.max_pool2d(encoder_block(sd, e + ".1", h), 2)
    h = F.max_pool2d(encoder_block(sd, e + ".3", h), 2)
    h = F.max_pool2d(encoder_block(sd, e + ".5", h), 2)
    return torch.relu(encoder_block(sd, e + ".7", h))


def vq_nearest(z: torch.Tensor, codebook: torch.Tensor) -> torch.Tensor:
    """vqvae_model.py:8-25: dist = (|c|^2 + |x|^2) - 2 x.c^T through addmm, first minimum wins.
    z [..., D] -> int64 [...]."""
    flat = z.reshape(-1, codebook.shape[1])
    c2 = (codebook ** 2).sum(dim=1)
    x2 = (flat ** 2).sum(dim=1, keepdim=True)
    dist = torch.addmm(c2 + x2, flat, codebook.t(), alpha=-2.0, beta=1.0)
    return dist.min(dim=1)[1].view(z.shape[:-1])


def vq_distances(z: torch.Tensor, codebook: torch.Tensor) -> torch.Tensor:
    flat = z.reshape(-1, codebook.shape[1])
    c2 = (codebook ** 2).sum(dim=1)
    x2 = (flat ** 2).sum(dim=1, keepdim=True)
    return torch.addmm(c2 + x2, flat, codebook.t(), alpha=-2.0, beta=1.0)


def vqvae_encode(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """VectorQuantizedVAE.encode (vqvae_model.py:233-237) + VQEmbedding.forward (:93-96)."""
    z = vqvae_encoder(sd, p, x).permute(0, 2, 3, 1).contiguous()
    return vq_nearest(z, sd[p + "codebook.embedding.weight"])


def vqvae_decode(sd: SD, p: str, ids: torch.Tensor) -> torch.Tensor:
    """VectorQuantizedVAE.decode (vqvae_model.py:239-242).  f4 decoder :180-189, f8 :203-214."""
    z = sd[p + "codebook.embedding.weight"][ids].permute(0, 3, 1, 2).contiguous()
    d = p + "decoder"
    if vqvae_down_ratio(sd, p) == 4:
        h = resblock(sd, d + ".1", resblock(sd, d + ".0", z))
        h = F.conv_transpose2d(torch.relu(h), sd[d + ".3.weight"], sd[d + ".3.bias"], stride=2, padding=1)
        h = torch.relu(_bn_eval(sd, d + ".4", h))
        h = F.conv_transpose2d(h, sd[d + ".6.weight"], sd[d + ".6.bias"], stride=2, padding=1)
        return torch.tanh(h)
    h = F.interpolate(decoder_block(sd, d + ".0", z), scale_factor=2, mode="nearest")
    h = F.interpolate(decoder_block(sd, d + ".2", h), scale_factor=2, mode="nearest")
    h = F.interpolate(decoder_block(sd, d + ".4", h), scale_factor=2, mode="nearest")
    h = torch.relu(decoder_block(sd, d + ".6", h))
    return torch.tanh(F.conv2d(h, sd[d + ".8.weight"], sd[d + ".8.bias"]))


# ----------------------------------------------------------------------------- attention
def mha(sd: SD, p: str, q_in: torch.Tensor, kv_in: torch.Tensor, n_head: int,
        attn_mask: Optional[torch.Tensor] = None, key_padding: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.MultiheadAttention forward, batch-first here: q_in [R, Tq, E], kv_in [R, Tk, E].
    attn_mask [Tq, Tk] additive; key_padding [R, Tk] bool (True = ignore)."""
    E = q_in.shape[-1]
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    hd = E // n_head
    q = F.linear(q_in, w[:E], b[:E])
    k = F.linear(kv_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(kv_in, w[2 * E:], b[2 * E:])
    R, Tq, Tk = q.shape[0], q.shape[1], k.shape[1]
    q = q.view(R, Tq, n_head, hd).transpose(1, 2)
    k = k.view(R, Tk, n_head, hd).transpose(1, 2)
    v = v.view(R, Tk, n_head, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    if attn_mask is not None:
        s = s + attn_mask
    if key_padding is not None:
        s = s.masked_fill(key_padding[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(R, Tq, E)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """mage_model.py:11-13."""
    return x * torch.sigmoid(1.702 * x)


def _ln(sd: SD, p: str, x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _mlp(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(quick_gelu(F.linear(x, sd[p + ".c_fc.weight"], sd[p + ".c_fc.bias"])),
                    sd[p + ".c_proj.weight"], sd[p + ".c_proj.bias"])


def text_encoder(sd: SD, p: str, text: torch.Tensor, padding_idx: int = 0) -> torch.Tensor:
    """TransformerTextEncoder.forward (mage_model.py:223-250) -> [B, S, out].
    Embedding LayerNorm eps is 1e-8 (:204); encoder layers are post-norm with erf-GELU
    (:192-200); keys beyond each caption's length are masked (:237,243) but padded
    *query* rows are still computed and returned non-zero."""
    B, S = text.shape
    width = sd[p + "token_embedding.weight"].shape[1]
    n_head = width // 32
    x = sd[p + "token_embedding.weight"][text] + sd[p + "positions.weight"][:S][None]
    x = _ln(sd, p + "layer_norm", x, 1e-8)
    x = x * (text != padding_idx).unsqueeze(-1).to(x.dtype)
    length = (text != padding_idx).sum(-1)
    key_pad = torch.arange(1, S + 1)[None, :] > length[:, None]
    li = 0
    while (p + f"transformer.layers.{li}.linear1.weight") in sd:
        lp = p + f"transformer.layers.{li}"
        x = _ln(sd, lp + ".norm1", x + mha(sd, lp + ".self_attn", x, x, n_head, key_padding=key_pad))
        ff = F.linear(F.gelu(F.linear(x, sd[lp + ".linear1.weight"], sd[lp + ".linear1.bias"])),
                      sd[lp + ".linear2.weight"], sd[lp + ".linear2.bias"])
        x = _ln(sd, lp + ".norm2", x + ff)
        li += 1
    x = _ln(sd, p + "ln_text_final", x)
    return F.linear(x, sd[p + "text_projection.weight"], sd[p + "text_projection.bias"])


def ma_encoder(sd: SD, p: str, q: torch.Tensor, kv: torch.Tensor, mage_plus: bool = False) -> torch.Tensor:
    """MAEncoder.forward (mage_model.py:114-117) over TransformerBlock.forward (:91-95).
    q [B, HW, C], kv [B, S, C].  MAGE: no ln_q/ln_kv, no key mask (:92).  MAGE+ (:93)
    applies ln_q / ln_kv (key mask is still None at both call sites :596,:657)."""
    li = 0
    x = q
    while (p + f"blocks.{li}.ln_2.weight") in sd:
        bp = p + f"blocks.{li}"
        n_head = x.shape[-1] // 32
        if mage_plus:
            x = x + mha(sd, bp + ".attn", _ln(sd, bp + ".ln_q", x), _ln(sd, bp + ".ln_kv", kv), n_head)
        else:
            x = x + mha(sd, bp + ".attn", x, kv, n_head)
        x = x + _mlp(sd, bp + ".mlp", _ln(sd, bp + ".ln_2", x))
        li += 1
    return x


def axial_block(sd: SD, p: str, x: torch.Tensor, axis: int, causal: bool) -> torch.Tensor:
    """AxialAttentionBlock.forward (mage_model.py:35-53) on x [B, L, H, W, C] without the
    permute/contiguous round trip: attention runs along ``axis`` (1=L, 2=H, 3=W)."""
    C = x.shape[-1]
    xt = x.movedim(axis, -2)
    lead = xt.shape[:-2]
    rows = xt.reshape(-1, xt.shape[-2], C)
    A = rows.shape[1]
    mask = torch.full((A, A), float("-inf")).triu_(1) if causal else None      # :367-372
    rows = rows + mha(sd, p + ".attn", _ln(sd, p + ".ln_1", rows), _ln(sd, p + ".ln_1", rows), C // 32, attn_mask=mask)
    rows = rows + _mlp(sd, p + ".mlp", _ln(sd, p + ".ln_2", rows))
    return rows.view(*lead, A, C).movedim(-2, axis).contiguous()


def flat_axial_decoder(sd: SD, p: str, motion: torch.Tensor, imgs: torch.Tensor) -> torch.Tensor:
    """FlatAxialDecoder.forward (mage_model.py:374-390), use_cids=True head.
    motion [B,H,W,Cc], imgs [B,L-1,H,W,Ci] -> logits [B,L-1,H,W,K]."""
    x = torch.cat([F.linear(motion, sd[p + "context_linear.weight"], sd[p + "context_linear.bias"]).unsqueeze(1),
                   F.linear(imgs, sd[p + "in_linear.weight"], sd[p + "in_linear.bias"])], 1)
    x = x + sd[p + "T_positional_embedding"]
    i = 0
    while (p + f"blocks.{i}.ln_1.weight") in sd:
        x = axial_block(sd, p + f"blocks.{i}", x, axis=i % 3 + 1, causal=(i % 3 == 0))    # :344,:382
        i += 1
    return F.linear(x[:, 1:], sd[p + "out.weight"], sd[p + "out.bias"])


# ----------------------------------------------------------------------------- MAGE
def _frame_features(sd: SD, tokens: torch.Tensor) -> torch.Tensor:
    """tokens int64 [B,T,h,w] -> conv3x3(embedding) + H/W positional tables, [B,T,h,w,C]
    (mage_model.py:581,586-588 / 644,648-649 / 674-676)."""
    B, T, h, w = tokens.shape
    emb = sd["visual_token_embedding.weight"][tokens].permute(0, 1, 4, 2, 3).reshape(B * T, -1, h, w)
    f = F.conv2d(emb, sd["conv.0.weight"], None, padding=1).view(B, T, -1, h, w).permute(0, 1, 3, 4, 2)
    return f + sd["H_positional_embedding"] + sd["W_positional_embedding"]


def adain(sd: SD, motion: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """Sampling-time randomness branch (mage_model.py:660-664, ADAIN2D :299-314).
    motion [B,H,W,C]; noise [B,64,H,W] is injected (the reference draws torch.randn)."""
    y = F.conv2d(noise, sd["conv_d2.weight"], None, padding=1)
    # nn.InstanceNorm2d(affine=False) written out (biased variance, eps 1e-5): F.instance_norm's CPU BACKWARD is wrong for a batch
    # of one (checked against the closed form and against autograd of this expression), and the training tests differentiate
    # through the oracle; the forward values are the same
    mp = motion.permute(0, 3, 1, 2).contiguous()                          # :606
    mean = mp.mean(dim=(2, 3), keepdim=True)
    var = mp.var(dim=(2, 3), unbiased=False, keepdim=True)
    m = (mp - mean) / torch.sqrt(var + 1e-5)
    gam = F.conv2d(F.conv2d(y, sd["adain.conv_mu.0.weight"], sd["adain.conv_mu.0.bias"], padding=1),
                   sd["adain.conv_mu.1.weight"], sd["adain.conv_mu.1.bias"], padding=1)
    bet = F.conv2d(F.conv2d(y, sd["adain.conv_var.0.weight"], sd["adain.conv_var.0.bias"], padding=1),
                   sd["adain.conv_var.1.weight"], sd["adain.conv_var.1.bias"], padding=1)
    return (gam * m + bet).permute(0, 2, 3, 1).contiguous()


def motion_anchor(sd: SD, tok0: torch.Tensor, text: torch.Tensor, speed: Optional[torch.Tensor],
                  noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Prologue of both forward and generate (mage_model.py:586-613 / 648-668)."""
    B, h, w = tok0.shape
    first = _frame_features(sd, tok0[:, None])[:, 0].reshape(B, h * w, -1)
    txt = text_encoder(sd, "text_encoder.", text)
    ma = ma_encoder(sd, "ma_encoder.", first, txt).view(B, h, w, -1)
    if noise is not None:
        ma = adain(sd, ma, noise)
    if speed is not None:
        ma = ma + (speed.view(B, 1) @ sd["speed_embedding"])[:, None, None, :]
    return ma


def mage_generate(sd: SD, batch: Dict[str, torch.Tensor], frames_length: int, noise: Optional[torch.Tensor] = None,
                  return_trace: bool = False):
    """MAGE.autoregressive_generate (mage_model.py:641-693), use_cids=True, with the
    reference's full recompute of the decoder in each of the L-1 iterations."""
    images = batch["images"]
    B = images.shape[0]
    fs = "first_stage_model."
    tok0 = vqvae_encode(sd, fs, images[:, 0])                       # :642
    ma = motion_anchor(sd, tok0, batch["text"], batch.get("speed"), noise)
    Lm1 = frames_length - 1
    cur = tok0[:, None].repeat(1, Lm1, 1, 1)                        # :670 future slots hold frame 0
    trace: List[torch.Tensor] = []
    logits = None
    for i in range(Lm1):                                            # :673-684
        logits = flat_axial_decoder(sd, "generate_model.", ma, _frame_features(sd, cur))
        if return_trace:
            trace.append(logits[:, i].clone())
        if i != Lm1 - 1:
            cur[:, i + 1] = logits[:, i].max(-1)[1]                 # :681-682
    gen_tokens = logits.max(-1)[1]                                  # :687
    frames = vqvae_decode(sd, fs, gen_tokens.view(B * Lm1, *gen_tokens.shape[2:]))
    frames = frames.view(B, Lm1, *frames.shape[1:])
    video = torch.cat([images[:, 0:1], frames], 1)                  # :691
    if return_trace:
        return video, gen_tokens, tok0, torch.stack(trace, 1)
    return video


def mage_forward_loss(sd: SD, batch: Dict[str, torch.Tensor], frames_length: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """MAGE.forward (mage_model.py:575-639), use_cids=True, randomness=False: teacher-forced
    logits and the cross-entropy over the K codes (:618).  Returns (loss, logits)."""
    images = batch["images"]
    B, L = images.shape[:2]
    tok = vqvae_encode(sd, "first_stage_model.", images.reshape(B * L, *images.shape[2:]))
    tok = tok.view(B, L, *tok.shape[1:])
    ma = motion_anchor(sd, tok[:, 0], batch["text"], batch.get("speed"))
    logits = flat_axial_decoder(sd, "generate_model.", ma, _frame_features(sd, tok[:, :frames_length - 1]))
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), tok[:, 1:frames_length].reshape(-1))
    return loss, logits

# ----------------------------------------------------------------------------- MAGE.forward with randomness=True
def basic_block(sd: SD, p: str, x: torch.Tensor, stride_t: int = 2) -> torch.Tensor:
    """BasicBlock.forward (mage_model.py:280-297): Conv3d 3x3x3 (temporal stride) -> GroupNorm(16) -> ReLU -> Conv3d ->
    GroupNorm, plus the Conv3d + GroupNorm downsample of the input, ReLU of the sum.  x [B,C,T,H,W]."""
    out = F.conv3d(x, sd[p + "conv1.weight"], None, stride=(stride_t, 1, 1), padding=1)
    out = F.relu(F.group_norm(out, 16, sd[p + "bn1.weight"], sd[p + "bn1.bias"]))
    out = F.conv3d(out, sd[p + "conv2.weight"], None, padding=1)
    out = F.group_norm(out, 16, sd[p + "bn2.weight"], sd[p + "bn2.bias"])
    res = F.conv3d(x, sd[p + "downsample.0.weight"], None, stride=(stride_t, 1, 1), padding=1)
    res = F.group_norm(res, 16, sd[p + "downsample.1.weight"], sd[p + "downsample.1.bias"])
    return F.relu(out + res)


def video_prior(sd: SD, x_emb: torch.Tensor) -> torch.Tensor:
    """self.conv3d (mage_model.py:496-501,602-603): four temporal-stride-2 BasicBlocks over the token embeddings of ALL frames,
    [B,L,C,h,w] -> [B,C,h,w] (L in 9..16 collapses to one frame; squeeze(2))."""
    v = x_emb.permute(0, 2, 1, 3, 4).contiguous()                        # :602
    for i in range(4):
        v = basic_block(sd, f"conv3d.{i}.", v, 2)
    return v.squeeze(2)


def mage_forward_loss_random(sd: SD, batch: Dict[str, torch.Tensor], frames_length: int, eps: torch.Tensor, alpha: float,
                             beta: float, auto_beta: bool = False, v_kl: float = 0.0, pid=None, test_noise: Optional[torch.Tensor] = None):
    """MAGE.forward (mage_model.py:575-639), use_cids=True, randomness=True.  eps [B,64,h,w] is the reparameterisation noise
    the reference draws with torch.randn_like (:571), injected.  Returns (final_loss, dict of the reference's loss_dict
    values without the train/val prefix, logits, video_emb_prior)."""
    images = batch["images"]
    B, L = images.shape[:2]
    tok = vqvae_encode(sd, "first_stage_model.", images.reshape(B * L, *images.shape[2:]))
    tok = tok.view(B, L, *tok.shape[1:])
    h, w = tok.shape[2:]
    x_emb = sd["visual_token_embedding.weight"][tok].permute(0, 1, 4, 2, 3)                       # :581
    prior = video_prior(sd, x_emb)                                                               # :602-603
    mu = F.conv2d(prior, sd["conv_mu2.weight"], sd["conv_mu2.bias"], padding=1)                  # :570
    logvar = F.conv2d(prior, sd["conv_var2.weight"], sd["conv_var2.bias"], padding=1)
    video_emb = eps * (0.5 * logvar).exp() + mu                                                  # :571-573
    if test_noise is not None:                                                                   # test_flag=True (:604-605), noise injected
        video_emb = test_noise
    speed = batch.get("speed")
    first = _frame_features(sd, tok[:, :1])[:, 0].reshape(B, h * w, -1)
    txt = text_encoder(sd, "text_encoder.", batch["text"])
    ma = ma_encoder(sd, "ma_encoder.", first, txt).view(B, h, w, -1)
    ma = adain(sd, ma, video_emb)                                                                # :606-609 (conv_d2 inside)
    speed_emb = None
    if speed is not None:
        speed_emb = speed.view(B, 1) @ sd["speed_embedding"]
        ma = ma + speed_emb[:, None, None, :]
    logits = flat_axial_decoder(sd, "generate_model.", ma, _frame_features(sd, tok[:, :frames_length - 1]))
    recon = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), tok[:, 1:frames_length].reshape(-1))
    mu2, lv2 = mu.reshape(B, -1), logvar.reshape(B, -1)
    kl = -0.5 * torch.mean(torch.sum(1 + lv2 - mu2.pow(2) - lv2.exp(), dim=1))                   # :623
    parts = {"prediction": recon.item(), "kl_loss": kl.item()}
    if auto_beta:
        beta, _ = pid.pid(v_kl, kl.item())                                                       # :627
        parts["beta"] = beta
        final = recon + beta * kl
    else:
        l2 = torch.mean(torch.pow(torch.norm(speed_emb, dim=-1), 2))                             # :631
        final = recon + beta * kl + alpha * l2
    parts["final_loss"] = final.item()
    return final, parts, logits, prior


# ----------------------------------------------------------------------------- MAGE+ (use_cids=False), sampling side
def _frame_features_latent(sd: SD, lat: torch.Tensor) -> torch.Tensor:
    """latents [B,T,E,h,w] -> Linear(E->C) -> conv3x3 + positional tables, [B,T,h,w,C] (mage_model.py:583,646,674-676)."""
    B, T, E, h, w = lat.shape
    emb = F.linear(lat.permute(0, 1, 3, 4, 2), sd["visual_token_embedding.weight"], sd["visual_token_embedding.bias"])
    emb = emb.permute(0, 1, 4, 2, 3).reshape(B * T, -1, h, w)
    f = F.conv2d(emb, sd["conv.0.weight"], None, padding=1).view(B, T, -1, h, w).permute(0, 1, 3, 4, 2)
    return f + sd["H_positional_embedding"] + sd["W_positional_embedding"]


def flat_axial_decoder_latent(sd: SD, p: str, motion: torch.Tensor, imgs: torch.Tensor) -> torch.Tensor:
    """FlatAxialDecoder.forward with the MAGE+ head (mage_model.py:350-354,387-388): GroupNorm(32) over
    [C, L-1, h, w] per clip -> SiLU -> Conv3d 1x1x1.  -> [B, L-1, h, w, out]."""
    x = torch.cat([F.linear(motion, sd[p + "context_linear.weight"], sd[p + "context_linear.bias"]).unsqueeze(1),
                   F.linear(imgs, sd[p + "in_linear.weight"], sd[p + "in_linear.bias"])], 1)
    x = x + sd[p + "T_positional_embedding"]
    i = 0
    while (p + f"blocks.{i}.ln_1.weight") in sd:
        x = axial_block(sd, p + f"blocks.{i}", x, axis=i % 3 + 1, causal=(i % 3 == 0))
        i += 1
    y = x[:, 1:].permute(0, 4, 1, 2, 3).contiguous()                     # :386
    y = F.silu(F.group_norm(y, 32, sd[p + "out.0.weight"], sd[p + "out.0.bias"], 1e-5))
    return F.conv3d(y, sd[p + "out.2.weight"], sd[p + "out.2.bias"]).permute(0, 2, 3, 4, 1)


def mage_generate_latent(sd: SD, batch: Dict[str, torch.Tensor], frames_length: int, lat0: torch.Tensor,
                         noise: Optional[torch.Tensor] = None, mage_plus: bool = False, return_motion: bool = False):
    """MAGE.autoregressive_generate for use_cids=False between the first stage's encode and decode:
    lat0 [B, E, h, w] (latents of frame 0) -> predicted latents [B, L-1, h, w, E]."""
    B, E, h, w = lat0.shape
    Lm1 = frames_length - 1
    first = _frame_features_latent(sd, lat0[:, None])[:, 0].reshape(B, h * w, -1)
    txt = text_encoder(sd, "text_encoder.", batch["text"])
    ma = ma_encoder(sd, "ma_encoder.", first, txt, mage_plus=mage_plus).view(B, h, w, -1)
    if noise is not None:
        ma = adain(sd, ma, noise)
    if batch.get("speed") is not None:
        ma = ma + (batch["speed"].view(B, 1) @ sd["speed_embedding"])[:, None, None, :]
    cur = lat0[:, None].repeat(1, Lm1, 1, 1, 1)
    pred = None
    for i in range(Lm1):
        pred = flat_axial_decoder_latent(sd, "generate_model.", ma, _frame_features_latent(sd, cur))
        if i != Lm1 - 1:
            cur[:, i + 1] = pred[:, i].permute(0, 3, 1, 2)
    return (pred, ma) if return_motion else pred


def mage_forward_loss_latent(sd: SD, batch: Dict[str, torch.Tensor], frames_length: int, lat: torch.Tensor, eps: torch.Tensor,
                             v_kl: float, pid, mage_plus: bool = False):
    """MAGE.forward (mage_model.py:575-639) for use_cids=False with randomness=True and auto_beta=True (config/mage+_*.yaml),
    between the external first stage's encode and the loss: lat [B,L,E,h,w] = first_stage_encode(images), eps the injected
    reparameterisation noise.  Returns (final_loss, parts, predicted latents [B,L-1,h,w,E])."""
    B, L, E, h, w = lat.shape
    x_emb = F.linear(lat.permute(0, 1, 3, 4, 2), sd["visual_token_embedding.weight"], sd["visual_token_embedding.bias"])
    x_emb = x_emb.permute(0, 1, 4, 2, 3)                                                          # :583
    prior = video_prior(sd, x_emb)
    mu = F.conv2d(prior, sd["conv_mu2.weight"], sd["conv_mu2.bias"], padding=1)
    logvar = F.conv2d(prior, sd["conv_var2.weight"], sd["conv_var2.bias"], padding=1)
    video_emb = eps * (0.5 * logvar).exp() + mu
    first = _frame_features_latent(sd, lat[:, :1])[:, 0].reshape(B, h * w, -1)
    txt = text_encoder(sd, "text_encoder.", batch["text"])
    ma = adain(sd, ma_encoder(sd, "ma_encoder.", first, txt, mage_plus=mage_plus).view(B, h, w, -1), video_emb)
    if batch.get("speed") is not None:
        ma = ma + (batch["speed"].view(B, 1) @ sd["speed_embedding"])[:, None, None, :]
    pred = flat_axial_decoder_latent(sd, "generate_model.", ma, _frame_features_latent(sd, lat[:, :frames_length - 1]))
    recon = F.mse_loss(pred.permute(0, 1, 4, 2, 3), lat[:, 1:frames_length])                       # :620
    mu2, lv2 = mu.reshape(B, -1), logvar.reshape(B, -1)
    kl = -0.5 * torch.mean(torch.sum(1 + lv2 - mu2.pow(2) - lv2.exp(), dim=1))
    beta, _ = pid.pid(v_kl, kl.item())
    final = recon + beta * kl
    return final, {"prediction": recon.item(), "kl_loss": kl.item(), "beta": beta, "final_loss": final.item()}, pred



# ----------------------------------------------------------------------------- stage-1 training (train_vqvae.py:13-35)
def _bn_train(sd: SD, p: str, x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """nn.BatchNorm2d in training mode: batch statistics (the running buffers are not touched here)."""
    return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, eps)


def _resblock_train(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ResBlock.forward (vqvae_model.py:111-124) with training-mode BatchNorm; the leading ReLU is in place, so the skip carries
    relu(x) too."""
    r = F.relu(x)
    y = F.conv2d(r, sd[p + ".block.1.weight"], sd[p + ".block.1.bias"], padding=1)
    y = F.relu(_bn_train(sd, p + ".block.2", y))
    y = F.conv2d(y, sd[p + ".block.4.weight"], sd[p + ".block.4.bias"])
    return r + _bn_train(sd, p + ".block.5", y)


def vqvae_train_forward(sd: SD, p: str, x: torch.Tensor):
    """VectorQuantizedVAE.forward of the f4 model in TRAINING mode (vqvae_model.py:244-248 over :171-190, straight_through :98-108,
    VectorQuantizationStraightThrough :34-65), differentiable w.r.t. the tensors of sd: returns (x_tilde, z_e_x, z_q_x) where the
    decoder saw the straight-through quantisation (gradient of its input goes to z_e_x) and z_q_x = codebook[ids] carries the
    codebook gradient."""
    h = F.conv2d(x, sd[p + "encoder.0.weight"], sd[p + "encoder.0.bias"], stride=2, padding=1)
    h = F.relu(_bn_train(sd, p + "encoder.1", h))
    h = F.conv2d(h, sd[p + "encoder.3.weight"], sd[p + "encoder.3.bias"], stride=2, padding=1)
    h = _resblock_train(sd, p + "encoder.4", h)
    z_e = _resblock_train(sd, p + "encoder.5", h)
    cb = sd[p + "codebook.embedding.weight"]
    zr = z_e.permute(0, 2, 3, 1).contiguous()
    with torch.no_grad():
        ids = vq_nearest(zr, cb)
    codes = cb[ids.reshape(-1)].view_as(zr)
    z_q_st = (zr + (codes.detach() - zr).detach()).permute(0, 3, 1, 2)      # value = codes, gradient -> z_e (straight-through)
    z_q = codes.permute(0, 3, 1, 2)
    d = _resblock_train(sd, p + "decoder.0", z_q_st)
    d = F.relu(_resblock_train(sd, p + "decoder.1", d))
    d = F.conv_transpose2d(d, sd[p + "decoder.3.weight"], sd[p + "decoder.3.bias"], stride=2, padding=1)
    d = F.relu(_bn_train(sd, p + "decoder.4", d))
    x_tilde = torch.tanh(F.conv_transpose2d(d, sd[p + "decoder.6.weight"], sd[p + "decoder.6.bias"], stride=2, padding=1))
    return x_tilde, z_e, z_q


def _block8_train(sd: SD, p: str, x: torch.Tensor, ks) -> torch.Tensor:
    """EncoderBlock / DecoderBlock.forward (vqvae_model.py:126-165): id_path(x) + block(x), ReLUs out of place; ks = kernel sizes."""
    idp = F.conv2d(x, sd[p + ".id_path.weight"], sd[p + ".id_path.bias"]) if (p + ".id_path.weight") in sd else x
    h = x
    for j, k in zip((1, 3, 5, 7), ks):
        h = F.conv2d(F.relu(h), sd[f"{p}.block.{j}.weight"], sd[f"{p}.block.{j}.bias"], padding=k // 2)
    return idp + h


def vqvae8_train_forward(sd: SD, p: str, x: torch.Tensor):
    """VectorQuantizedVAE.forward of the f8 (CATER) model (vqvae_model.py:244-248 over :192-214; no BatchNorm, so train() and eval()
    compute the same values), differentiable w.r.t. the tensors of sd, straight-through quantiser as in vqvae_train_forward."""
    h = F.conv2d(x, sd[p + "encoder.0.weight"], sd[p + "encoder.0.bias"], padding=3)
    for bi in (1, 3, 5, 7):
        h = _block8_train(sd, p + f"encoder.{bi}", h, (3, 3, 3, 1))
        if bi != 7:
            h = F.max_pool2d(h, 2)
    z_e = F.relu(h)
    cb = sd[p + "codebook.embedding.weight"]
    zr = z_e.permute(0, 2, 3, 1).contiguous()
    with torch.no_grad():
        ids = vq_nearest(zr, cb)
    codes = cb[ids.reshape(-1)].view_as(zr)
    z_q_st = (zr + (codes.detach() - zr).detach()).permute(0, 3, 1, 2)
    z_q = codes.permute(0, 3, 1, 2)
    d = z_q_st
    for bi in (0, 2, 4, 6):
        d = _block8_train(sd, p + f"decoder.{bi}", d, (1, 3, 3, 3))
        if bi != 6:
            d = F.interpolate(d, scale_factor=2, mode="nearest")
    x_tilde = torch.tanh(F.conv2d(F.relu(d), sd[p + "decoder.8.weight"], sd[p + "decoder.8.bias"]))
    return x_tilde, z_e, z_q


def vqvae_train_loss(sd: SD, p: str, x: torch.Tensor, beta: float = 2.0):
    """The objective of train_vqvae.py:20-27 (reconstruction + vector-quantisation + beta * commitment)."""
    f8 = (p + "encoder.7.block.1.weight") in sd
    x_tilde, z_e, z_q = (vqvae8_train_forward if f8 else vqvae_train_forward)(sd, p, x)
    rec, vql, com = F.mse_loss(x_tilde, x), F.mse_loss(z_q, z_e.detach()), F.mse_loss(z_e, z_q.detach())
    return rec + vql + beta * com, (rec, vql, com), (x_tilde, z_e, z_q)
