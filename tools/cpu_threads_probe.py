"""Calibrate the torch thread count for the CPU baseline on the GPU box's host (oversubscription is slow)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config
from oracle import mage_oracle as O
L = 4
m = instantiate_from_config(synth.mnist_model_config(frames_length=L)).eval()
synth.fill_state_dict(m, 0)
sd = {k: v.detach() for k, v in m.state_dict().items()}
batch = synth.synth_batch_mnist(2, L, seed=1)
print("cpu_count", os.cpu_count())
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        O.mage_generate(sd, batch, L)
        t0 = time.perf_counter(); O.mage_generate(sd, batch, L); dt = time.perf_counter() - t0
    print(f"threads {th}: {dt:.2f} s  ({2*L/dt:.2f} frames/s)", flush=True)
