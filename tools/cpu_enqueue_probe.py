"""How far ahead of the GPU is the Python launch loop?  Time for autoregressive_generate() to RETURN (all launches enqueued)
vs time until the GPU has finished.  Tuning only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config
dev = torch.device("cuda", 0)
B, L = 64, 16
model = instantiate_from_config(synth.mnist_model_config(frames_length=L)).eval()
synth.fill_state_dict(model, 0)
model = model.to(dev).set_precision("bf16")
batch = {k: v.to(dev) for k, v in synth.synth_batch_mnist(B, L, seed=100).items()}
for mode in ("full", "incremental"):
    model.ar_mode = mode
    for _ in range(2): model.autoregressive_generate(batch)
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter(); model.autoregressive_generate(batch); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{mode}: enqueue returned after {1e3 * (t1 - t0):.1f} ms, GPU done after {1e3 * (t2 - t0):.1f} ms")
