#!/usr/bin/env python
"""Per-parameter gradient errors of the randomness-branch training path against float64 autograd through the oracle.  Tuning only."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth
from oracle import mage_oracle as O
from tests.helpers import build_mage, cpu_sd

B, L, beta, alpha = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
seed = 41
cfg = synth.cater_model_config(frames_length=L, width=64, layers=3, vq_dim=32, K=64)
cfg["params"]["beta"], cfg["params"]["alpha"] = beta, alpha
m = build_mage(cfg, seed, "cuda:0")
batch = synth.synth_batch_cater(B, L, seed=seed, text_len=9)
eps = torch.randn(B, 64, 16, 16, generator=torch.Generator().manual_seed(seed))
sd = {k: (v.double().requires_grad_() if v.is_floating_point() and not k.startswith("first_stage_model.") else (v.double() if v.is_floating_point() else v))
      for k, v in cpu_sd(m).items()}
b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
loss, parts, _, _ = O.mage_forward_loss_random(sd, b64, L, eps.double(), alpha=alpha, beta=beta)
names = [k for k, v in sd.items() if v.requires_grad]
gs = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)))
db = {k: v.to("cuda:0") for k, v in batch.items()}
db["reparam_noise"] = eps.to("cuda:0")
l2, ld = m(db)
print("loss", l2.item(), loss.item(), ld)
l2.backward()
rows = []
for n, p in m.named_parameters():
    if p.grad is None or gs.get(n) is None:
        continue
    g = gs[n]
    own = g.abs().max().item()
    err = (p.grad.double().cpu() - g).abs().max().item()
    rows.append((err / max(own, 1e-300), n, own, err))
for r in sorted(rows, reverse=True)[:int(sys.argv[5]) if len(sys.argv) > 5 else 25]:
    print("%-55s rel %.3e  own max %.3e  abs err %.3e" % (r[1], r[0], r[2], r[3]))
