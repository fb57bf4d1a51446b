#!/usr/bin/env python
"""The reference's training loop (main_mage.py:139-154) on the HIP path: DataLoader with the MovingMnist batch contract ->
model(batch) -> loss.backward() -> optimizer.step(), timed per step.  usage: train_probe.py [B] [L] [precision] [steps] [mnist|cater|magep]"""
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from mage_amd.optim import FlatAdam  # noqa: E402
from mage_amd.utils import glue, synth  # noqa: E402
from mage_amd.utils.util import instantiate_from_config  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = "cuda:0"
family = sys.argv[5] if len(sys.argv) > 5 else "mnist"              # "cater": config/mage_caterv1.yaml's model (f8 VQ-VAE, randomness branch)
cfg = {"mnist": synth.mnist_model_config, "cater": synth.cater_model_config, "magep": synth.magep_model_config}[family](frames_length=L)
model = instantiate_from_config(cfg)                                 # "magep": config/mage+_caterv2.yaml's MAGE side over the stand-in latent first stage
synth.fill_state_dict(model, 0)
model = model.to(dev).set_precision(prec).train()
opt = FlatAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6)
if family == "mnist":
    ds = glue.SyntheticMovingMnist(B * (steps + 1), frames_length=L)
    loader = torch.utils.data.DataLoader(ds, batch_size=B, shuffle=False, collate_fn=ds.collate_fn, num_workers=0)
else:
    loader = [synth.synth_batch_cater(B, L, seed=200 + i, **({"vocab": 50} if family == "magep" else {})) for i in range(steps + 1)]
times, losses = [], []
# the synthetic dataset draws its clips on the host (~4 ms per clip, one process): with num_workers=0 the GPU idles ~250 ms between
# steps, drops its clocks, and the next forward pass measures the ramp (46 vs 133 ms for the same kernels).  A real input pipeline
# keeps batches ahead of the device; here they are produced through the same DataLoader / collate contract before the timed loop.
batches = [{k: v.to(dev) for k, v in batch.items()} for batch in loader]
for it, batch in enumerate(batches):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad()
    loss, ld = model(batch)
    t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    times.append((t1 - t0, t2 - t1, t3 - t2))
    losses.append(loss.item())
    print(f"iter {it}: train_loss = {losses[-1]:.6f}  forward {1e3 * (t1 - t0):.1f} ms  backward {1e3 * (t2 - t1):.1f} ms  step {1e3 * (t3 - t2):.2f} ms", flush=True)
f, b, s = (sum(t[i] for t in times[2:]) / len(times[2:]) for i in range(3))
ntok = B * L * 256
print(f"B={B} L={L} {prec}: forward {1e3 * f:.1f} ms, backward {1e3 * b:.1f} ms, optimizer {1e3 * s:.2f} ms per step -> {B * L / (f + b + s):.0f} frames/s trained, "
      f"{3 * 38.96e6 * ntok / (f + b + s) / 1e12:.0f} TFLOP/s (3x forward decoder FLOPs), peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; "
      f"loss {losses[0]:.4f} -> {losses[-1]:.4f}")
