#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE (read-only, /root/reference)
on portable synthetic weights + inputs.  Build container only: the reference never
travels to the GPU box; only the small fixtures written here do.

    python tools/gen_golden.py            # regenerate everything (about a minute)

Every fixture stores inputs that cannot be regenerated from mage_amd.utils.synth,
the reference's outputs (tokens, logit slices, frames), fp64 checksums of big
tensors and per-position top-2 margins, so that a consumer can tell an argmin /
argmax near-tie from a real mismatch.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mage_amd.utils import synth  # noqa: E402
from tools._ref_import import import_reference, to_cfg  # noqa: E402

OUT = os.environ.get("MAGE_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")      # tools/check_goldens.sh regenerates into a temp dir
torch.set_num_threads(8)
torch.manual_seed(0)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def top2_margin(x, largest=True):
    v = x.topk(2, dim=-1, largest=largest)[0]
    return (v[..., 0] - v[..., 1]).abs().float()


def chk(x):
    x = x.double()
    return np.array([x.sum().item(), x.abs().sum().item(), (x * x).sum().item()], np.float64)


def build_ref_mage(ref_mage, cfg, seed):
    p = cfg["params"]
    width, layers = p["vision_width"], p["generate_decoder_config"]["params"]["layers"]
    m = ref_mage.MAGE(**to_cfg(p)).eval()
    synth.fill_state_dict(m, seed, d_model=width, n_layers=layers)
    return m


def gen_forward_random(ref_mage):
    # ---- 7c. MAGE.forward with randomness=True: Conv3d video prior, reparameterisation (noise injected), KL + l2 terms ----
    print("mage_cater_forward_small")
    cfg = synth.cater_model_config(frames_length=10, width=64, layers=3, vq_dim=32, K=64)
    m = build_ref_mage(ref_mage, cfg, 61)
    batch = synth.synth_batch_cater(2, 10, seed=61, text_len=12)
    eps = torch.from_numpy(synth.rng_for(61, "reparam_noise").standard_normal((2, 64, 16, 16)).astype(np.float32))
    real_randn_like = torch.randn_like

    def fake_randn_like(t, *a, **k):
        return eps.clone() if tuple(t.shape) == tuple(eps.shape) else real_randn_like(t, *a, **k)
    cap = {}
    h1 = m.conv3d.register_forward_hook(lambda mod, i, o: cap.__setitem__("prior", o.detach().clone()))
    h2 = m.generate_model.register_forward_hook(lambda mod, i, o: cap.__setitem__("logits", o.detach().clone()))
    torch.randn_like = fake_randn_like
    try:
        with torch.no_grad():
            loss, ld = m({k: v.clone() for k, v in batch.items()})
    finally:
        torch.randn_like = real_randn_like
        h1.remove(); h2.remove()
    prior = cap["prior"].squeeze(2)
    save("mage_cater_forward_small", seed=61, B=2, L=10, width=64, layers=3, vq_dim=32, K=64, text_len=12, eps=eps,
         final_loss=np.float64(loss.item()), prediction=np.float64(ld["val/prediction"]), kl_loss=np.float64(ld["val/kl_loss"]),
         prior_sub=prior[:, ::4, ::2, ::2].contiguous(), prior_chk=chk(prior), logits_sub=cap["logits"][:, ::3, ::4, ::4, ::8].contiguous(),
         logits_chk=chk(cap["logits"]))
    # MAGE+ (use_cids=False, auto_beta): MSE on the latents of a stand-in first stage, PID-controlled beta
    print("mage_plus_forward_small")
    cfg = synth.magep_model_config(frames_length=10, width=64, layers=3)
    m = build_ref_mage(ref_mage, cfg, 71)
    batch = synth.synth_batch_cater(2, 10, seed=71, text_len=12, vocab=50)
    eps2 = torch.from_numpy(synth.rng_for(71, "reparam_noise").standard_normal((2, 64, 16, 16)).astype(np.float32))

    def fake_randn_like2(t, *a, **k):
        return eps2.clone() if tuple(t.shape) == tuple(eps2.shape) else real_randn_like(t, *a, **k)
    cap = {}
    h2 = m.generate_model.register_forward_hook(lambda mod, i, o: cap.__setitem__("pred", o.detach().clone()))
    torch.randn_like = fake_randn_like2
    try:
        with torch.no_grad():
            loss, ld = m({k: v.clone() for k, v in batch.items()})
    finally:
        torch.randn_like = real_randn_like
        h2.remove()
    save("mage_plus_forward_small", seed=71, B=2, L=10, width=64, layers=3, text_len=12, eps=eps2,
         final_loss=np.float64(loss.item()), prediction=np.float64(ld["val/prediction"]), kl_loss=np.float64(ld["val/kl_loss"]),
         beta=np.float64(ld["val/beta"]), pred=cap["pred"])



def _inject_randn(noise):
    real_randn = torch.randn

    def fake_randn(*a, **k):
        shape = a[0] if len(a) == 1 and isinstance(a[0], (list, tuple)) else a
        return noise.clone() if tuple(shape) == tuple(noise.shape) else real_randn(*a, **k)
    return real_randn, fake_randn


def gen_cater_fullwidth(ref_mage):
    """cfg4's MODEL at full width (config/mage_caterv1.yaml: d=512, 6 blocks, f8 VQ-VAE dim 256 -> codebook D=1024, K=512,
    randomness=True) on a short clip (B=1, L=4): pins the kernel dispatch the full-size CATER runs use (256-row tiles, K=1024
    quantiser, f8 stack at dim 256) at the 1e-4 gate."""
    print("mage_cater_fullwidth")
    from oracle import mage_oracle as O
    L, B, seed = 4, 1, 43
    cfg = synth.cater_model_config(frames_length=L)
    m = build_ref_mage(ref_mage, cfg, seed)
    batch = synth.synth_batch_cater(B, L, seed=seed, text_len=14)
    noise = torch.from_numpy(synth.rng_for(seed, "video_noise").standard_normal((B, 64, 16, 16)).astype(np.float32))
    real_randn, fake_randn = _inject_randn(noise)
    trace = []
    orig_gen = m.generate_model.forward

    def spy(motion, imgs):
        out = orig_gen(motion, imgs)
        trace.append((motion.clone(), out.clone()))
        return out
    m.generate_model.forward = spy
    torch.randn = fake_randn
    try:
        with torch.no_grad():
            video = m.autoregressive_generate({k: v.clone() for k, v in batch.items()})
            x0 = batch["images"][:, 0].contiguous()
            z_e = m.first_stage_model.encoder(x0)
            tok0 = m.first_stage_model.encode(x0)
            dist = O.vq_distances(z_e.permute(0, 2, 3, 1).contiguous(), m.first_stage_model.codebook.embedding.weight)
    finally:
        torch.randn = real_randn
    step_logits = torch.stack([trace[i][1][:, i] for i in range(L - 1)], 1)
    save("mage_cater_fullwidth", seed=seed, B=B, L=L, text_len=14, noise=noise, tok0=tok0.to(torch.int16),
         tok0_margin=top2_margin(dist, largest=False).view(tok0.shape), z_e_slice=z_e[:, :8], z_e_chk=chk(z_e),
         motion_sub=trace[0][0][:, ::4, ::4].contiguous(), motion_chk=chk(trace[0][0]),
         gen_tokens=trace[-1][1].max(-1)[1].to(torch.int16), margin=top2_margin(step_logits),
         step_logits_sub=step_logits[:, :, ::4, ::4].contiguous(), step_logits_chk=chk(step_logits),
         video_sub=video[..., ::4, ::4].contiguous(), video_chk=chk(video))


def gen_mage_plus_block(ref_mage):
    """MAGE+ with the TransformerBlock variant the reference's comment prescribes for it (mage_model.py:93: ln_q / ln_kv
    applied, residual from the un-normalised q).  The shipped source has :92 active; here the block's forward is replaced, in
    this process only, by the documented MAGE+ line -- the reference tree is untouched."""
    print("mage_plus_block_small")

    def fwd93(self, q, k, v, key_mask=None, need_weights=False):
        x = q + self.dropout(self.attention(self.ln_q(q), self.ln_kv(k), self.ln_kv(v), key_mask))
        x = x + self.dropout(self.mlp(self.ln_2(x)))
        return x
    orig_fwd = ref_mage.TransformerBlock.forward
    ref_mage.TransformerBlock.forward = fwd93
    real_randn = torch.randn
    real_randn_like = torch.randn_like
    try:
        cfg = synth.magep_model_config(frames_length=4, width=64, layers=3)
        m = build_ref_mage(ref_mage, cfg, 52)
        batch = synth.synth_batch_cater(2, 4, seed=52, text_len=12, vocab=50)
        noise = torch.from_numpy(synth.rng_for(52, "video_noise").standard_normal((2, 64, 16, 16)).astype(np.float32))
        _, fake_randn = _inject_randn(noise)
        trace = []
        orig_gen = m.generate_model.forward

        def spyp(motion, imgs):
            out = orig_gen(motion, imgs)
            trace.append((motion.clone(), out.clone()))
            return out
        m.generate_model.forward = spyp
        torch.randn = fake_randn
        with torch.no_grad():
            video = m.autoregressive_generate({k: v.clone() for k, v in batch.items()})
        torch.randn = real_randn
        m.generate_model.forward = orig_gen
        # and the teacher-forced loss of the same variant (L=10 so the Conv3d prior collapses to one frame)
        cfg10 = synth.magep_model_config(frames_length=10, width=64, layers=3)
        m10 = build_ref_mage(ref_mage, cfg10, 72)
        batch10 = synth.synth_batch_cater(2, 10, seed=72, text_len=12, vocab=50)
        eps = torch.from_numpy(synth.rng_for(72, "reparam_noise").standard_normal((2, 64, 16, 16)).astype(np.float32))
        torch.randn_like = lambda t, *a, **k: eps.clone() if tuple(t.shape) == tuple(eps.shape) else real_randn_like(t, *a, **k)
        cap = {}
        h = m10.generate_model.register_forward_hook(lambda mod, i, o: cap.__setitem__("pred", o.detach().clone()))
        with torch.no_grad():
            loss, ld = m10({k: v.clone() for k, v in batch10.items()})
        h.remove()
    finally:
        torch.randn = real_randn
        torch.randn_like = real_randn_like
        ref_mage.TransformerBlock.forward = orig_fwd
    save("mage_plus_block_small", seed=52, B=2, L=4, width=64, layers=3, text_len=12, noise=noise, motion=trace[0][0],
         pred_latents=trace[-1][1], video_sub=video[..., ::4, ::4].contiguous(), video_chk=chk(video),
         fwd_seed=72, fwd_L=10, fwd_eps=eps, fwd_final_loss=np.float64(loss.item()), fwd_prediction=np.float64(ld["val/prediction"]),
         fwd_kl_loss=np.float64(ld["val/kl_loss"]), fwd_beta=np.float64(ld["val/beta"]), fwd_pred_sub=cap["pred"][:, ::3].contiguous())



def gen_vqvae_train(ref_vq):
    """Stage-1 training step of the reference itself (train_vqvae.py:13-27): VectorQuantizedVAE in train() mode (BatchNorm on
    batch statistics, straight-through quantiser), the three-term loss, loss.backward(): loss values, every parameter's gradient
    (dim 32: in full; dim 256: checksums + slices) and the BatchNorm running buffers after the step."""
    import torch.nn.functional as F
    cases = (("vqvae_f4_train_small", 4, 32, 64, 4, 13, True), ("vqvae_f4_train", 4, 256, 512, 2, 14, False),
             ("vqvae_f8_train_small", 8, 32, 64, 2, 15, True), ("vqvae_f8_train", 8, 128, 256, 1, 16, False))
    if "--only-vqvae8-train" in sys.argv:
        cases = cases[2:]
    for tag, ratio, dim, K, n_img, seed, full in cases:
        print(tag)
        m = ref_vq.VectorQuantizedVAE(1 if ratio == 4 else 3, ratio, dim, K)
        synth.fill_state_dict(m, seed)
        m.train()
        if ratio == 4:
            x = synth.synth_batch_mnist(n_img, 1, seed=seed)["images"][:, 0].contiguous()
        else:                                                  # CATER-like RGB frames, 64x64 (any multiple of 8 works; train_vqvae.py:97)
            x = synth.synth_batch_cater(n_img, 1, seed=seed, res=64)["images"][:, 0].contiguous()
        x_tilde, z_e, z_q = m(x.clone())
        rec, vql, com = F.mse_loss(x_tilde, x), F.mse_loss(z_q, z_e.detach()), F.mse_loss(z_e, z_q.detach())
        loss = rec + vql + 2.0 * com
        loss.backward()
        out = dict(seed=seed, dim=dim, K=K, n_img=n_img, ratio=ratio, beta=2.0, loss=np.float64(loss.item()), rec=np.float64(rec.item()),
                   vq=np.float64(vql.item()), commit=np.float64(com.item()), x_tilde_chk=chk(x_tilde.detach()), z_e_chk=chk(z_e.detach()),
                   z_q_chk=chk(z_q.detach()), x_tilde_sub=x_tilde.detach()[:, :, ::4, ::4].contiguous())
        names = []
        for n, p_ in m.named_parameters():
            names.append(n)
            g = p_.grad if p_.grad is not None else torch.zeros_like(p_)
            out["gchk." + n] = chk(g)
            out["gmax." + n] = np.float64(g.abs().max().item())
            if full or g.numel() <= 4096:
                out["g." + n] = g.clone()
            else:
                out["gs." + n] = g.flatten()[::max(1, g.numel() // 1024)][:1024].clone()
        for n, b in m.named_buffers():
            if n.endswith("running_mean") or n.endswith("running_var"):
                out["buf." + n] = b.clone()
        out["param_names"] = np.array(names)
        save(tag, **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_mage, ref_vq = import_reference()
    if "--only-forward-random" in sys.argv:      # regenerate just fixture 7c
        gen_forward_random(ref_mage)
        return
    if "--only-round2" in sys.argv:              # the fixtures added in round 2
        gen_cater_fullwidth(ref_mage)
        gen_mage_plus_block(ref_mage)
        gen_vqvae_train(ref_vq)
        return
    if "--only-vqvae-train" in sys.argv or "--only-vqvae8-train" in sys.argv:
        gen_vqvae_train(ref_vq)
        return

    # ---- 1. VQ unit: exact ties, near ties, reference-init regime --------------------------
    print("vq unit")
    g = synth.rng_for(7, "vq_unit")
    K, D = 32, 16
    cb = g.standard_normal((K, D)).astype(np.float32)
    cb[5] = cb[2]                      # duplicate code: exact tie, first index (2) must win
    cb[17] = cb[9]
    z = g.standard_normal((96, D)).astype(np.float32)
    z[0], z[1] = cb[2], cb[9]          # inputs sitting on duplicated codes
    z[2] = 0.5 * (cb[3] + cb[4])       # equidistant in exact arithmetic
    z[3] = 0.0
    idx = ref_vq.vq(torch.from_numpy(z), torch.from_numpy(cb))
    # reference-init regime: tiny codebook U(-1/K, 1/K) vs O(1) inputs (vqvae_model.py:91)
    cb2 = (g.random((64, 32)) * 2 - 1).astype(np.float32) / 64
    z2 = g.standard_normal((4, 8, 8, 32)).astype(np.float32)
    idx2 = ref_vq.vq(torch.from_numpy(z2), torch.from_numpy(cb2))
    from oracle import mage_oracle as O
    d2 = O.vq_distances(torch.from_numpy(z2), torch.from_numpy(cb2))
    save("vq_unit", cb=cb, z=z, idx=idx, cb2=cb2, z2=z2, idx2=idx2, margin2=top2_margin(d2, largest=False))

    # ---- 2. f4 VQ-VAE (MNIST) ---------------------------------------------------------------
    print("vqvae f4")
    vq4 = ref_vq.VectorQuantizedVAE(1, 4, 256, 512).eval()
    synth.fill_state_dict(vq4, 11)
    x = synth.synth_batch_mnist(3, 2, seed=11)["images"][:, :, :].reshape(6, 1, 64, 64)[:4].contiguous()
    with torch.no_grad():
        z_e = vq4.encoder(x.clone())
        ids = vq4.encode(x.clone())
        dist = O.vq_distances(z_e.permute(0, 2, 3, 1).contiguous(), vq4.codebook.embedding.weight)
        rec = vq4.decode(ids)
        x_tilde, z_e2, z_q = vq4(x.clone())
    save("vqvae_f4", seed=11, x=x, ids=ids.to(torch.int16), z_e_slice=z_e[:, :8], z_e_chk=chk(z_e),
         margin=top2_margin(dist, largest=False).view(ids.shape), rec=rec, rec_chk=chk(rec),
         fwd_x_tilde_chk=chk(x_tilde), z_q_chk=chk(z_q))

    # ---- 3. f8 VQ-VAE (CATER) at reduced dim -------------------------------------------------
    print("vqvae f8")
    vq8 = ref_vq.VectorQuantizedVAE(3, 8, 32, 64).eval()
    synth.fill_state_dict(vq8, 12)
    x8 = synth.synth_batch_cater(2, 1, seed=12)["images"][:, 0].contiguous()
    with torch.no_grad():
        z_e8 = vq8.encoder(x8)
        ids8 = vq8.encode(x8)
        dist8 = O.vq_distances(z_e8.permute(0, 2, 3, 1).contiguous(), vq8.codebook.embedding.weight)
        rec8 = vq8.decode(ids8)
    save("vqvae_f8", seed=12, dim=32, K=64, ids=ids8.to(torch.int16), z_e_slice=z_e8[:, :8], z_e_chk=chk(z_e8),
         margin=top2_margin(dist8, largest=False).view(ids8.shape), rec_sub=rec8[..., ::4, ::4].contiguous(),
         rec_chk=chk(rec8))

    # ---- 4. MAGE, MNIST cfg at full width, short clip: every stage ----------------------------
    for tag, B, L, seed, ragged, digits, tl in (("mage_mnist_L4", 2, 4, 21, False, 1, 11),
                                                ("mage_mnist_L6_ragged", 3, 6, 22, True, 2, 20)):
        print(tag)
        cfg = synth.mnist_model_config(frames_length=L)
        m = build_ref_mage(ref_mage, cfg, seed)
        batch = synth.synth_batch_mnist(B, L, seed=seed, digits=digits, text_len=tl, ragged_text=ragged)
        trace = []
        orig_gen = m.generate_model.forward

        def spy(motion, imgs, _o=orig_gen, _t=trace):
            out = _o(motion, imgs)
            _t.append((motion.clone(), out.clone()))
            return out
        m.generate_model.forward = spy
        with torch.no_grad():
            txt = m.text_encoder(batch["text"])
            video = m.autoregressive_generate({k: v.clone() for k, v in batch.items()})
            tok0 = m.first_stage_encode(batch["images"][:, 0:1])[:, 0]
        motion = trace[0][0]
        step_logits = torch.stack([trace[i][1][:, i] for i in range(L - 1)], 1)      # [B,L-1,h,w,K]
        gen_tok = trace[-1][1].max(-1)[1]
        m.generate_model.forward = orig_gen
        with torch.no_grad():
            loss, ld = m({k: v.clone() for k, v in batch.items()})
        save(tag, seed=seed, B=B, L=L, digits=digits, text_len=tl, ragged=ragged,
             text=batch["text"], text_emb=txt, tok0=tok0.to(torch.int16), motion_sub=motion[:, ::4, ::4].contiguous(),
             motion_chk=chk(motion), gen_tokens=gen_tok.to(torch.int16),
             step_logits_sub=step_logits[:, :, ::4, ::4].contiguous(), step_logits_chk=chk(step_logits),
             margin=top2_margin(step_logits), video=video, loss=np.float64(loss.item()),
             loss_dict_keys=np.array(sorted(ld.keys())))

    # ---- 5. cfg1 shape (L=16) tokens only ------------------------------------------------------
    print("mage_mnist_L16")
    cfg = synth.mnist_model_config(frames_length=16)
    m = build_ref_mage(ref_mage, cfg, 0)
    batch = synth.synth_batch_mnist(2, 16, seed=0)
    trace = []
    orig_gen = m.generate_model.forward

    def spy16(motion, imgs):
        out = orig_gen(motion, imgs)
        i = len(trace)
        trace.append(out[:, i].clone())
        spy16.last = out
        return out
    m.generate_model.forward = spy16
    with torch.no_grad():
        video = m.autoregressive_generate({k: v.clone() for k, v in batch.items()})
    step_logits = torch.stack(trace, 1)
    save("mage_mnist_L16", seed=0, B=2, L=16, gen_tokens=spy16.last.max(-1)[1].to(torch.int16),
         margin=top2_margin(step_logits), step_logits_sub=step_logits[:, :, ::8, ::8, ::4].contiguous(),
         video_chk=chk(video), video_sub=video[:, :, :, ::2, ::2].contiguous())

    # ---- 6. reduced-width model (d=64, 2 heads), exercises non-512 shapes ----------------------
    print("mage_small_d64")
    cfg = synth.mnist_model_config(frames_length=5, width=64, layers=3, vq_dim=32, K=64)
    m = build_ref_mage(ref_mage, cfg, 31)
    batch = synth.synth_batch_mnist(3, 5, seed=31, text_len=9, ragged_text=True)
    trace = []
    orig_gen = m.generate_model.forward

    def spy64(motion, imgs):
        out = orig_gen(motion, imgs)
        trace.append(out.clone())
        return out
    m.generate_model.forward = spy64
    with torch.no_grad():
        video = m.autoregressive_generate({k: v.clone() for k, v in batch.items()})
    step_logits = torch.stack([trace[i][:, i] for i in range(4)], 1)
    save("mage_small_d64", seed=31, B=3, L=5, width=64, layers=3, vq_dim=32, K=64, text_len=9,
         gen_tokens=trace[-1].max(-1)[1].to(torch.int16), margin=top2_margin(step_logits),
         step_logits=step_logits, video=video)

    # ---- 7. cfg4 shape at reduced size: f8 first stage + randomness (ADAIN) with injected noise -
    print("mage_cater_small")
    cfg = synth.cater_model_config(frames_length=4, width=64, layers=3, vq_dim=32, K=64)
    m = build_ref_mage(ref_mage, cfg, 41)
    batch = synth.synth_batch_cater(2, 4, seed=41, text_len=12)
    noise = torch.from_numpy(synth.rng_for(41, "video_noise").standard_normal((2, 64, 16, 16)).astype(np.float32))
    real_randn = torch.randn

    def fake_randn(*a, **k):
        shape = a[0] if len(a) == 1 and isinstance(a[0], (list, tuple)) else a
        if tuple(shape) == tuple(noise.shape):
            return noise.clone()
        return real_randn(*a, **k)
    trace = []
    orig_gen = m.generate_model.forward

    def spyc(motion, imgs):
        out = orig_gen(motion, imgs)
        trace.append((motion.clone(), out.clone()))
        return out
    m.generate_model.forward = spyc
    torch.randn = fake_randn
    try:
        with torch.no_grad():
            video = m.autoregressive_generate({k: v.clone() for k, v in batch.items()})
    finally:
        torch.randn = real_randn
    step_logits = torch.stack([trace[i][1][:, i] for i in range(3)], 1)
    save("mage_cater_small", seed=41, B=2, L=4, width=64, layers=3, vq_dim=32, K=64, text_len=12, noise=noise,
         motion=trace[0][0], gen_tokens=trace[-1][1].max(-1)[1].to(torch.int16), margin=top2_margin(step_logits),
         step_logits=step_logits, video_sub=video[..., ::4, ::4].contiguous(), video_chk=chk(video))
    # ---- 7b. MAGE+ side (use_cids=False, GroupNorm/SiLU/Conv3d head) over a stand-in latent first stage -------------
    print("mage_plus_small")
    cfg = synth.magep_model_config(frames_length=4, width=64, layers=3)
    m = build_ref_mage(ref_mage, cfg, 51)
    batch = synth.synth_batch_cater(2, 4, seed=51, text_len=12, vocab=50)
    noise = torch.from_numpy(synth.rng_for(51, "video_noise").standard_normal((2, 64, 16, 16)).astype(np.float32))

    def fake_randn2(*a, **k):
        shape = a[0] if len(a) == 1 and isinstance(a[0], (list, tuple)) else a
        return noise.clone() if tuple(shape) == tuple(noise.shape) else real_randn(*a, **k)
    trace = []
    orig_gen = m.generate_model.forward

    def spyp(motion, imgs):
        out = orig_gen(motion, imgs)
        trace.append((motion.clone(), out.clone()))
        return out
    m.generate_model.forward = spyp
    torch.randn = fake_randn2
    try:
        with torch.no_grad():
            video = m.autoregressive_generate({k: v.clone() for k, v in batch.items()})
    finally:
        torch.randn = real_randn
    save("mage_plus_small", seed=51, B=2, L=4, width=64, layers=3, text_len=12, noise=noise, motion=trace[0][0],
         pred_latents=trace[-1][1], pred_step0=trace[0][1], video_sub=video[..., ::4, ::4].contiguous(), video_chk=chk(video))

    gen_forward_random(ref_mage)
    gen_cater_fullwidth(ref_mage)
    gen_mage_plus_block(ref_mage)
    gen_vqvae_train(ref_vq)
    # ---- 8. state_dict layout (keys, shapes, dtypes) of the reference modules: the drop-in boundary ----------------
    import json
    layout = {}
    for tag, cfg in (("mnist_L16", synth.mnist_model_config(frames_length=16)), ("caterv1_L10", synth.cater_model_config(frames_length=10)),
                     ("magep_caterv2_L10", synth.magep_model_config(frames_length=10))):
        rm = ref_mage.MAGE(**to_cfg(cfg["params"]))
        layout[tag] = [[k, list(v.shape), str(v.dtype)] for k, v in rm.state_dict().items()]
    json.dump(layout, open(os.path.join(OUT, "state_dict_layout.json"), "w"))
    print("  wrote state_dict_layout.json")
    print("done")


if __name__ == "__main__":
    main()
