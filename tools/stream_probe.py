#!/usr/bin/env python
"""bf16 mode with x kept in bf16 between the blocks (default) vs the fp32 stream + bf16 copy (stream_bf16 = False): free-running token
agreement with the f16x3 (fp32-class) run at cfg2, teacher-forced logits against fp32 mode, and the time per call."""
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth  # noqa: E402
from mage_amd.utils.util import instantiate_from_config  # noqa: E402

dev = "cuda:0"
m = instantiate_from_config(synth.mnist_model_config(frames_length=16)).eval()
synth.fill_state_dict(m, 0)
m = m.to(dev)
batch = {k: v.to(dev) for k, v in synth.synth_batch_mnist(64, 16, seed=100).items()}
m.set_precision("f16x3")
m.autoregressive_generate(batch)
ref = m.last_tokens.clone()
small = {k: v[:8] for k, v in batch.items()}
m.set_precision("fp32")
_, lg32 = m.teacher_forced_logits(small)
for sb in (True, False):
    m.set_precision("bf16")
    m.generate_model.stream_bf16 = sb
    for mode in ("full", "incremental"):
        m.ar_mode = mode
        m.autoregressive_generate(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m.autoregressive_generate(batch)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        eq = (m.last_tokens == ref).float()
        print(f"stream_bf16={sb!s:5} {mode:11}: {ms:7.2f} ms per call | tokens equal to the f16x3 run: first generated frame "
              f"{eq[:, 0].mean():.4f}, second {eq[:, 1].mean():.4f}, all positions {eq.mean():.4f}")
    m.ar_mode = "full"
    _, lg = m.teacher_forced_logits(small)
    err = (lg - lg32).abs()
    agree = (lg.argmax(-1) == lg32.argmax(-1)).float().mean().item()
    print(f"stream_bf16={sb!s:5} teacher-forced vs fp32 mode (8 clips): max |dlogit| {err.max().item():.4f}, mean {err.mean().item():.5f}, "
          f"argmax agreement {agree:.4f}")
