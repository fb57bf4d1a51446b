"""Incremental / full AR loop at cfg2 / cfg4 sizes, eager launches vs HIP-graph replay of the whole call (MAGE.use_graph).  Tuning only."""
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
    for mode in ("incremental", "full"):
        m.ar_mode = mode
        ref = None
        for ug in (False, True):
            m.use_graph = ug
            m.autoregressive_generate(batch); m.autoregressive_generate(batch)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 5 if mode == "incremental" else 2
            for _ in range(n): m.autoregressive_generate(batch)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
            if ref is None: ref = m.last_tokens.clone()
            print(f"{name} {mode:11} bf16 B={B}: use_graph={ug!s:5}: {ms:8.2f} ms per call ({m.last_call_mode}), tokens identical: {torch.equal(ref, m.last_tokens)}, "
                  f"mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    del m
    torch.cuda.empty_cache()
