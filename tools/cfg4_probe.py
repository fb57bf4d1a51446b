#!/usr/bin/env python
"""BASELINE cfg4 (CATER-GEN-v1 128x128, 32 frames, batch 32, mage_caterv1.yaml: f8 VQ-VAE, randomness branch) on one GPU: ms per call."""
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth  # noqa: E402
from mage_amd.utils.util import instantiate_from_config  # noqa: E402

dev = "cuda:0"
B, L = 32, 32
m = instantiate_from_config(synth.cater_model_config()).eval()
synth.fill_state_dict(m, 0)
m = m.to(dev)
batch = {k: v.to(dev) for k, v in synth.synth_batch_cater(B, L, seed=1).items()}
for prec in ("bf16", "f16x3"):
    m.set_precision(prec)
    for mode in ("full", "incremental"):
        m.ar_mode = mode
        m.autoregressive_generate(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 2
        for _ in range(n):
            m.autoregressive_generate(batch)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"cfg4 {prec:6} {mode:11}: {ms:8.1f} ms per call, {B * L / ms * 1e3:8.0f} frames/s (128x128, {L} frames, batch {B})")
