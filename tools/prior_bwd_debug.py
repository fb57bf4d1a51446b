#!/usr/bin/env python
"""Backward of the Conv3d video prior alone against float64 autograd through the oracle's video_prior.  Tuning only."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth
from mage_amd.modules import mage_train_prior as P
from oracle import mage_oracle as O
from tests.helpers import build_mage, cpu_sd

for B, L in [(1, 9), (2, 9), (2, 12), (1, 16), (3, 10)]:
    cfg = synth.cater_model_config(frames_length=L, width=64, layers=3, vq_dim=32, K=64)
    m = build_mage(cfg, 41, "cuda:0")
    g = torch.Generator().manual_seed(B * 100 + L)
    tok = torch.randint(0, 64, (B, L, 256), generator=g)
    if len(sys.argv) > 1:                                                    # the tokens of a synthetic CATER batch instead
        batch = synth.synth_batch_cater(B, L, seed=41, text_len=9)
        tok = m.first_stage_encode(batch["images"].to("cuda:0")).reshape(B, L, 256).cpu()
        print("   distinct tokens", tok.unique().numel())
    dprior = torch.randn(B * 256, 64, generator=g)
    sd = {k: v.double().requires_grad_() for k, v in cpu_sd(m).items() if k.startswith(("conv3d.", "visual_token_embedding"))}
    x_emb = sd["visual_token_embedding.weight"][tok.view(B, L, 16, 16)].permute(0, 1, 4, 2, 3)
    pr = O.video_prior(sd, x_emb)                                            # [B, C, h, w]
    loss = (pr.permute(0, 2, 3, 1).reshape(B * 256, 64) * dprior.double()).sum()
    names = list(sd)
    gs = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
    with torch.no_grad():
        blocks = []
        out = m._video_prior(tok.to("cuda:0"), tape=blocks)
        fe = (out.cpu().double() - pr.permute(0, 2, 3, 1).reshape(B * 256, 64)).abs().max().item()
        grads = {}
        d = m._derived.get(m._build)
        dxa, ds = P._prior_backward(m, d, blocks, dprior.to("cuda:0"), grads, B)
        emb = torch.zeros(64, 64, device="cuda:0")
        from mage_amd import ops
        ops.embedding_bwd(tok.reshape(-1).to("cuda:0"), dxa, emb, group=L * 256, group_stride=ds, off=256)
        grads["visual_token_embedding.weight"] = emb
    worst = sorted(((grads[k].double().cpu() - gs[k]).abs().max().item() / gs[k].abs().max().item(), k) for k in gs)[::-1]
    print(f"B={B} L={L}: forward err {fe:.2e}; worst:", ", ".join(f"{k} {r:.1e}" for r, k in worst[:4]), "| best:", f"{worst[-1][1]} {worst[-1][0]:.1e}")
