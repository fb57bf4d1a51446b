"""Every precision mode x AR mode x model family (MNIST f4, CATER f8 randomness, MAGE+ latent) at a tiny size, generation and one training step: each
combination either runs or raises its intended, explicit error.  A crash screen, not a parity test (those are tests/test_gpu_*.py)."""
import sys, torch, traceback
sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth
from tests.helpers import build_mage
DEV = "cuda:0"
def dev(b): return {k: v.to(DEV) for k, v in b.items()}
fam = {
 "mnist": (lambda L: synth.mnist_model_config(frames_length=L), lambda B, L: synth.synth_batch_mnist(B, L, seed=1)),
 "cater": (lambda L: synth.cater_model_config(frames_length=L), lambda B, L: dict(synth.synth_batch_cater(B, L, seed=1), video_noise=torch.randn(B, 64, 16, 16))),
 "magep": (lambda L: synth.magep_model_config(frames_length=L, width=64, layers=3), lambda B, L: dict(synth.synth_batch_cater(B, L, seed=1, text_len=8, vocab=50), video_noise=torch.randn(B, 64, 16, 16))),
}
for name, (cfg, mk) in fam.items():
    L, B = 4, 2
    try:
        m = build_mage(cfg(L), 0, DEV)
    except Exception as e:
        print(name, "BUILD FAILED", repr(e)[:200]); continue
    batch = dev(mk(B, L))
    for prec in ("fp32", "f16x3", "bf16x3", "bf16", "f16"):
        for mode in ("full", "incremental"):
            try:
                m.set_precision(prec); m.ar_mode = mode
                v = m.autoregressive_generate(batch)
                ok = bool(torch.isfinite(v).all())
                print(f"{name:6s} {prec:7s} {mode:11s} ok finite={ok}")
            except Exception as e:
                print(f"{name:6s} {prec:7s} {mode:11s} RAISED {type(e).__name__}: {str(e)[:160]}")
    # training forward / backward
    for prec in ("fp32", "bf16", "f16x3", "f16"):
        try:
            m.set_precision(prec); m.train()
            for p in m.parameters(): p.requires_grad_(True)
            if hasattr(m.first_stage_model, "requires_grad_"): m.first_stage_model.requires_grad_(False)
            out = m(batch)
            loss = out[0] if isinstance(out, (tuple, list)) else out
            loss = loss if torch.is_tensor(loss) and loss.dim() == 0 else (loss["loss"] if isinstance(loss, dict) else loss.sum())
            loss.backward()
            print(f"{name:6s} train {prec:6s} ok loss={float(loss):.4f}")
            m.zero_grad(); m.eval()
        except Exception as e:
            print(f"{name:6s} train {prec:6s} RAISED {type(e).__name__}: {str(e)[:160]}"); m.eval()
