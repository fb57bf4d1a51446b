"""VQ-VAE (MNIST f4) decode / encode timing with a per-launch breakdown.  Tuning only.
usage: python tools/bench_vqvae.py [frames=960] [chunk=<decode_chunk>] [precision=bf16]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import ops
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config

kv = dict(a.split("=") for a in sys.argv[1:])
N = int(kv.get("frames", 960)); prec = kv.get("precision", "bf16")
dev = torch.device("cuda", 0)
cfg = synth.mnist_model_config(frames_length=16)
model = instantiate_from_config(cfg).eval()
synth.fill_state_dict(model, 0)
vq = model.first_stage_model.to(dev)
vq.set_precision(prec)
if "chunk" in kv: vq.decode_chunk = int(kv["chunk"])
ids = torch.randint(0, 512, (N, 16, 16), device=dev)
x = torch.rand(N, 1, 64, 64, device=dev) * 2 - 1
for name, fn in (("decode", lambda: vq.decode(ids)), ("encode", lambda: vq.encode(x))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    ops.PROFILE.reset(enabled=True); fn(); prof = ops.PROFILE.summary(); ops.PROFILE.enabled = False
    tot = sum(v["ms"] for v in prof.values())
    print(f"{name}: {N} frames {prec} chunk {vq.decode_chunk}: {ms:.3f} ms  ({N / ms * 1e3:.0f} frames/s); profiled kernels {tot:.3f} ms")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        extra = f"{v['flops'] / v['ms'] / 1e9:8.1f} TFLOP/s" if v["flops"] else (f"{v['bytes'] / v['ms'] / 1e9:8.2f} TB/s" if v["bytes"] else "")
        print(f"   {k:40s} {v['calls']:3d} calls {v['ms']:8.3f} ms {extra}")
