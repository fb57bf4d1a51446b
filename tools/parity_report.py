#!/usr/bin/env python
"""Measured parity of the HIP path against every committed golden (GPU box): max |error| per stage, so that the gates in
tests/test_gpu_parity.py are chosen from numbers, and committed under profiles/ as evidence.

    python tools/parity_report.py [fp32|f16x3|bf16x3] > gpurun_out/parity_report.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mage_amd.utils import synth  # noqa: E402
from tests.helpers import build_mage as _build_mage, golden, t  # noqa: E402

DEV = "cuda:0"
PREC = sys.argv[1] if len(sys.argv) > 1 else "fp32"        # fp32 | f16x3 | bf16x3 (| bf16: free-running tokens will differ)


def build_mage(cfg, seed, device):
    return _build_mage(cfg, seed, device).set_precision(PREC)


def dev(b):
    return {k: v.to(DEV) for k, v in b.items()}


def mx(a, b):
    return (a.detach().float().cpu() - b.float()).abs().max().item()


def line(tag, **kv):
    print(f"{tag:28s} " + "  ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in kv.items()), flush=True)


def tok_report(got, want, margin):
    got, want, margin = got.cpu().numpy().reshape(-1), np.asarray(want).reshape(-1), np.asarray(margin).reshape(-1)
    bad = got != want
    return {"mismatch": int(bad.sum()), "of": int(bad.size), "min_margin": float(margin.min()),
            "max_margin_at_mismatch": float(margin[bad].max()) if bad.any() else 0.0}


def main():
    torch.manual_seed(0)
    print(f"precision mode: {PREC}", flush=True)
    for tag in ("mage_mnist_L4", "mage_mnist_L6_ragged"):
        g = golden(tag)
        B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
        m = build_mage(synth.mnist_model_config(frames_length=L), seed, DEV)
        db = dev(synth.synth_batch_mnist(B, L, seed=seed, digits=int(g["digits"]), text_len=int(g["text_len"]), ragged_text=bool(g["ragged"])))
        v = m.autoregressive_generate(db)
        line(tag, logits=mx(m.last_logits[:, :, ::4, ::4], t(g["step_logits_sub"])), video=mx(v, t(g["video"])),
             logit_absmax=float(np.abs(g["step_logits_sub"]).max()), **tok_report(m.last_tokens, g["gen_tokens"], g["margin"]))
    g = golden("mage_mnist_L16")
    m = build_mage(synth.mnist_model_config(frames_length=16), int(g["seed"]), DEV)
    v = m.autoregressive_generate(dev(synth.synth_batch_mnist(int(g["B"]), 16, seed=int(g["seed"]))))
    line("mage_mnist_L16", logits=mx(m.last_logits[:, :, ::8, ::8, ::4], t(g["step_logits_sub"])), video=mx(v[:, :, :, ::2, ::2], t(g["video_sub"])),
         **tok_report(m.last_tokens, g["gen_tokens"], g["margin"]))
    g = golden("mage_small_d64")
    m = build_mage(synth.mnist_model_config(frames_length=int(g["L"]), width=64, layers=3, vq_dim=32, K=64), int(g["seed"]), DEV)
    v = m.autoregressive_generate(dev(synth.synth_batch_mnist(int(g["B"]), int(g["L"]), seed=int(g["seed"]), text_len=int(g["text_len"]), ragged_text=True)))
    line("mage_small_d64", logits=mx(m.last_logits, t(g["step_logits"])), video=mx(v, t(g["video"])),
         logit_absmax=float(np.abs(g["step_logits"]).max()), **tok_report(m.last_tokens, g["gen_tokens"], g["margin"]))
    # ---- randomness branch (ADAIN), width 64: stage by stage
    g = golden("mage_cater_small")
    cfg = synth.cater_model_config(frames_length=int(g["L"]), width=64, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, int(g["seed"]), DEV)
    db = dev(synth.synth_batch_cater(int(g["B"]), int(g["L"]), seed=int(g["seed"]), text_len=int(g["text_len"])))
    db["video_noise"] = t(g["noise"]).to(DEV)
    B = int(g["B"])
    tok0 = m.first_stage_encode(db["images"][:, 0:1])[:, 0]
    ma = m._motion_anchor(tok0.reshape(B, -1), db, db["video_noise"]).view(B, 16, 16, -1)
    v = m.autoregressive_generate(db)
    line("mage_cater_small", motion=mx(ma, t(g["motion"])), motion_absmax=float(np.abs(g["motion"]).max()),
         logits=mx(m.last_logits, t(g["step_logits"])), logit_absmax=float(np.abs(g["step_logits"]).max()),
         video=mx(v[..., ::4, ::4], t(g["video_sub"])), **tok_report(m.last_tokens, g["gen_tokens"], g["margin"]))
    try:
        # where the error enters: the same anchor WITHOUT the ADAIN branch against the oracle in fp64
        from oracle import mage_oracle as O
        from tests.helpers import cpu_sd
        sd64 = {k: (x.double() if x.is_floating_point() else x) for k, x in cpu_sd(m).items()}
        cb = synth.synth_batch_cater(int(g["B"]), int(g["L"]), seed=int(g["seed"]), text_len=int(g["text_len"]))
        ma64 = O.motion_anchor(sd64, tok0.cpu().view(B, 16, 16), cb["text"], cb["speed"].double(), t(g["noise"]).double())
        ma64_plain = O.motion_anchor(sd64, tok0.cpu().view(B, 16, 16), cb["text"], cb["speed"].double(), None)
        ma_plain = m._motion_anchor.__func__(_NoRand(m), tok0.reshape(B, -1), db, None).view(B, 16, 16, -1)
        line("  vs fp64 oracle", motion_adain=mx(ma, ma64), motion_plain=mx(ma_plain, ma64_plain), ref_motion_vs_fp64=mx(t(g["motion"]), ma64))
        lg64 = O.flat_axial_decoder(sd64, "generate_model.", ma64, O._frame_features(sd64, torch.cat([tok0.cpu().view(B, 1, 16, 16), t(g["gen_tokens"]).long()[:, :-1]], 1)))
        line("  vs fp64 oracle", logits_tf=mx(m.last_logits, lg64), ref_logits_vs_fp64=mx(t(g["step_logits"]), lg64))
        # decoder alone fed the fp64 anchor rounded to fp32: isolates the decoder's own error from the anchor's
        feats = m._frame_features(torch.cat([tok0.view(B, 1, 256), t(g["gen_tokens"]).long().to(DEV).view(B, -1, 256)[:, :-1]], 1).contiguous(), torch.float32)
        lg_dec = m.generate_model._run(ma64.float().to(DEV).reshape(B * 256, -1).contiguous(), feats, B=B, hh=16, ww=16).view(B, -1, 16, 16, lg64.shape[-1])
        line("  decoder on exact anchor", logits=mx(lg_dec, lg64))
    except Exception as e:      # diagnostic only
        print('  fp64 stage report failed:', repr(e), flush=True)
    g = golden("mage_cater_fullwidth")
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    m = build_mage(synth.cater_model_config(frames_length=L), seed, DEV)
    db = dev(synth.synth_batch_cater(B, L, seed=seed, text_len=int(g["text_len"])))
    db["video_noise"] = t(g["noise"]).to(DEV)
    tok0 = m.first_stage_encode(db["images"][:, 0:1])[:, 0]
    ma = m._motion_anchor(tok0.reshape(B, -1), db, db["video_noise"]).view(B, 16, 16, -1)
    v = m.autoregressive_generate(db)
    line("mage_cater_fullwidth", motion=mx(ma[:, ::4, ::4], t(g["motion_sub"])), logits=mx(m.last_logits[:, :, ::4, ::4], t(g["step_logits_sub"])),
         video=mx(v[..., ::4, ::4], t(g["video_sub"])), tok0_mismatch=int((tok0.cpu() != t(g["tok0"]).long()).sum()),
         **tok_report(m.last_tokens, g["gen_tokens"], g["margin"]))
    for tag, shipped in ((("mage_plus_small", True), ("mage_plus_block_small", False)) if PREC != "f16" else ()):      # (f16: VQ-token path only)
        g = golden(tag)
        B, L = int(g["B"]), int(g["L"])
        m = build_mage(synth.magep_model_config(frames_length=L, width=64, layers=3), int(g["seed"]), DEV)
        if shipped:
            m.ma_encoder.mage_plus = False
        db = dev(synth.synth_batch_cater(B, L, seed=int(g["seed"]), text_len=int(g["text_len"]), vocab=50))
        db["video_noise"] = t(g["noise"]).to(DEV)
        v = m.autoregressive_generate(db)
        line(tag, pred_latents=mx(m.last_logits, t(g["pred_latents"])), latent_absmax=float(np.abs(g["pred_latents"]).max()),
             video=mx(v[..., ::4, ::4], t(g["video_sub"])))


class _NoRand:
    """View of a MAGE with the randomness branch switched off (for the stage-by-stage report only)."""

    def __init__(self, m):
        object.__setattr__(self, "_m", m)

    def __getattr__(self, k):
        if k == "randomness":
            return False
        return getattr(object.__getattribute__(self, "_m"), k)


if __name__ == "__main__":
    main()
