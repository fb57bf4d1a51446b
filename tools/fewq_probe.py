#!/usr/bin/env python
"""The incremental step's temporal attention (one query per sequence against the K,V cache) at every position: µs per launch."""
import sys

import torch

sys.path.insert(0, "/root/repo")
from mage_amd import ops  # noqa: E402

B, L, hw, Cc, H = 64, 16, 256, 512, 16
dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(0)
q = torch.randn(B * hw, Cc, generator=g).to(dev).to(torch.bfloat16)
kv = torch.randn(B * L * hw, 2 * Cc, generator=g).to(dev).to(torch.bfloat16)
ao = torch.empty(B * hw, Cc, device=dev, dtype=torch.bfloat16)


def run(nk):
    ops.attention(q, kv, kv[:, Cc:], ao, ldq=Cc, ldk=2 * Cc, ldv=2 * Cc, ldo=Cc, n_seq=B * hw, inner=hw, nq=1, nk=nk, n_head=H,
                  q_outer_stride=hw, q_axis_stride=hw, kv_outer_stride=L * hw, kv_axis_stride=hw, causal=True)


for nk in (1, 2, 4, 8, 12, 16):
    for _ in range(3):
        run(nk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run(nk)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = (B * hw * Cc * 2 * 2 + nk * B * hw * 2 * Cc * 2) / 1e6
    print(f"nk={nk:2d}: {us:7.1f} us   {mb:6.1f} MB -> {mb / us / 1e3 * 1e3:.2f} TB/s".replace("TB/s", "GB/ms"))
