"""Incremental AR loop at cfg2 / cfg4 sizes with the clips split into n groups on n HIP streams (MAGE.streams): do independent groups'
kernels fill each other's ramps and tails?  Tokens are compared with the one-stream run.  Tuning only."""
import sys, time
import torch
sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config

dev = "cuda:0"
for name, cfg, mk, B, L in (("cfg2", synth.mnist_model_config(frames_length=16), synth.synth_batch_mnist, 64, 16),
                            ("cfg4", synth.cater_model_config(frames_length=32), synth.synth_batch_cater, 32, 32)):
    m = instantiate_from_config(cfg).eval()
    synth.fill_state_dict(m, 0)
    m = m.to(dev).set_precision("bf16")
    b = mk(B, L, seed=3)
    if name == "cfg4":
        b["video_noise"] = torch.randn(B, 64, 16, 16, generator=torch.Generator().manual_seed(5))
    batch = {k: v.to(dev) for k, v in b.items()}
    m.ar_mode = "incremental"
    ref = None
    for n in (1, 2, 4):
        m.streams = n
        m.autoregressive_generate(batch); m.autoregressive_generate(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m.autoregressive_generate(batch)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        if ref is None: ref = m.last_tokens.clone()
        print(f"{name} incremental bf16 B={B}: streams={n}: {ms:7.2f} ms per call, tokens identical to one stream: {torch.equal(ref, m.last_tokens)}")
    del m
    torch.cuda.empty_cache()
