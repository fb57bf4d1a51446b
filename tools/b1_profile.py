#!/usr/bin/env python
"""B = 1 incremental calls (graph off) for rocprofv3 --kernel-trace: where one clip's 12 ms go."""
import sys, time
import torch
sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth  # noqa: E402
from mage_amd.utils.util import instantiate_from_config  # noqa: E402
m = instantiate_from_config(synth.mnist_model_config(frames_length=16)).eval()
synth.fill_state_dict(m, 0)
m = m.to("cuda:0").set_precision(__import__("os").environ.get("B1_PRECISION", "bf16"))
m.ar_mode = sys.argv[1] if len(sys.argv) > 1 else "incremental"
import os
m.use_graph = bool(os.environ.get("B1_GRAPH"))
batch = {k: v.to("cuda:0") for k, v in synth.synth_batch_mnist(1, 16, seed=100).items()}
for _ in range(2):
    m.autoregressive_generate(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    m.autoregressive_generate(batch)
torch.cuda.synchronize()
print(f"B=1 {m.ar_mode}: {(time.perf_counter() - t0) / 4 * 1e3:.2f} ms per call")
