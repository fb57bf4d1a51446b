#!/usr/bin/env python
"""Eager launch loop vs whole-call HIP-graph replay, both AR modes, cfg2 shape (GPU box)."""
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth  # noqa: E402
from mage_amd.utils.util import instantiate_from_config  # noqa: E402

B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 16
m = instantiate_from_config(synth.mnist_model_config(frames_length=L)).eval()
synth.fill_state_dict(m, 0)
m = m.to("cuda:0").set_precision("bf16")
batch = {k: v.to("cuda:0") for k, v in synth.synth_batch_mnist(B, L, seed=100).items()}


def timeit(n=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m.autoregressive_generate(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def enqueue_ms():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.autoregressive_generate(batch)
    return (time.perf_counter() - t0) * 1e3


for mode in ("incremental", "full"):
    m.ar_mode = mode
    m.use_graph = False
    m.autoregressive_generate(batch)
    tok = m.last_tokens.clone()
    e = timeit()
    m.use_graph = True
    m.autoregressive_generate(batch)
    t0 = time.perf_counter()
    m.autoregressive_generate(batch)
    torch.cuda.synchronize()
    cap = (time.perf_counter() - t0) * 1e3
    g = timeit()
    same = torch.equal(tok, m.last_tokens)
    print(f"{mode:12s} eager {e:8.2f} ms   graph replay {g:8.2f} ms   (capture+first replay {cap:.0f} ms)  mode={m.last_call_mode} tokens identical {same}  "
          f"mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
