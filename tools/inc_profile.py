#!/usr/bin/env python
"""A few incremental-mode calls, eager, for rocprofv3 --kernel-trace --stats: GPU-busy time per call vs wall."""
import os
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth  # noqa: E402
from mage_amd.utils.util import instantiate_from_config  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "incremental"
m = instantiate_from_config(synth.mnist_model_config(frames_length=16)).eval()
synth.fill_state_dict(m, 0)
m = m.to("cuda:0").set_precision(os.environ.get("INC_PRECISION", "bf16"))
m.ar_mode = mode
import os
m.use_graph = bool(os.environ.get("INC_GRAPH"))
m.streams = int(os.environ.get("INC_STREAMS", "1"))
m.autoregressive_generate({k: v.to("cuda:0") for k, v in synth.synth_batch_mnist(64, 16, seed=100).items()})
batch = {k: v.to("cuda:0") for k, v in synth.synth_batch_mnist(64, 16, seed=100).items()}
m.autoregressive_generate(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    m.autoregressive_generate(batch)
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t0) / 4 * 1e3:.2f} ms per call (5 calls traced incl. warm-up)")
