import sys
import torch
sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth
from tests.helpers import build_mage
DEV = "cuda:0"
cfg = synth.mnist_model_config(frames_length=4, width=64, layers=3, vq_dim=32, K=64)
m = build_mage(cfg, 37, DEV)
batch = {k: v.to(DEV) for k, v in synth.synth_batch_mnist(2, 4, seed=37).items()}


def loss_at():
    torch.manual_seed(99)
    return m(batch)[0]


def check(tag, pname, eps):
    m.zero_grad(set_to_none=True)
    loss_at().backward()
    p = dict(m.named_parameters())[pname]
    g = p.grad.clone()
    d = torch.randn(p.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    d = d / d.norm()
    def shift(a):
        with torch.no_grad():
            p.add_(a * d)
    shift(eps); lp = loss_at().item()
    shift(-2 * eps); lm = loss_at().item()
    shift(eps)
    print(f"{tag:28s} {pname:48s} eps {eps:.0e} fd {(lp - lm) / (2 * eps):+.6f} analytic {(g * d).sum().item():+.6f}", flush=True)


names = ["generate_model.blocks.1.mlp.c_fc.weight", "generate_model.out.weight", "generate_model.blocks.2.mlp.c_proj.weight",
         "generate_model.blocks.2.attn.out_proj.weight", "generate_model.blocks.0.attn.in_proj_weight", "ma_encoder.blocks.0.mlp.c_fc.weight",
         "text_encoder.transformer.layers.1.linear1.weight", "conv.0.weight"]
for mode in ("eval", "train"):
    m.train(mode == "train")
    for n in names:
        for eps in (2e-2, 5e-3):
            check(mode, n, eps)
