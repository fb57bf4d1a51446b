"""Bitwise screen of the cross-kernel invariant at odd batch sizes: the tokens of B = 1, 3, 5, 7, 33 clips (full and incremental loop) against the same rows of a
batch of 40, MNIST f4 (L = 16) and CATER f8 (L = 5), bf16 / f16 / f16x3 -- different batch sizes land on different GEMM kernels and tile shapes."""
import sys, torch
sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth
from tests.helpers import build_mage
DEV = "cuda:0"
for fam, cfg, mk, L in (("mnist", synth.mnist_model_config, synth.synth_batch_mnist, 16), ("cater", synth.cater_model_config, synth.synth_batch_cater, 5)):
    m = build_mage(cfg(frames_length=L), 0, DEV)
    big = mk(40, L, seed=2)
    if fam == "cater": big["video_noise"] = torch.randn(40, 64, 16, 16, generator=torch.Generator().manual_seed(3))
    big = {k: v.to(DEV) for k, v in big.items()}
    for prec in ("bf16", "f16", "f16x3"):
        m.set_precision(prec); m.ar_mode = "full"
        m.autoregressive_generate(big); ref = m.last_tokens.clone()
        for B in (1, 3, 5, 7, 33):
            sub = {k: v[:B] for k, v in big.items()}
            res = []
            for mode in ("full", "incremental"):
                m.ar_mode = mode
                try:
                    m.autoregressive_generate(sub); res.append(bool(torch.equal(m.last_tokens, ref[:B])))
                except Exception as e:
                    res.append(f"RAISED {type(e).__name__}: {str(e)[:100]}")
            print(fam, prec, "B =", B, "tokens == rows of the batch of 40 (full, incremental):", res)
