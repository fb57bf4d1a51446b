"""f8 VQ-VAE (CATER: 128x128 RGB, dim 256, codebook D = 1024) decode / encode timing (BASELINE cfg4's first stage).  Tuning only.
usage: python tools/bench_vqvae_f8.py [frames=992] [precision=bf16] [reps=5]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd import ops
from mage_amd.modules.vqvae_model import VectorQuantizedVAE
from mage_amd.utils import synth

kv = dict(a.split("=") for a in sys.argv[1:])
N = int(kv.get("frames", 992)); prec = kv.get("precision", "bf16"); reps = int(kv.get("reps", 5))
dev = torch.device("cuda", 0)
vq = VectorQuantizedVAE(3, 8, 256, 512).eval()
synth.fill_state_dict(vq, 0)
vq = vq.to(dev)
vq.set_precision(prec)
ids = torch.randint(0, 512, (N, 16, 16), device=dev)
x = torch.rand(N // 4, 3, 128, 128, device=dev) * 2 - 1
for name, fn, n, fl in (("decode", lambda: vq.decode(ids), N, 11.333e9), ("encode", lambda: vq.encode(x), N // 4, 14.185e9)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    ops.PROFILE.reset(enabled=True); fn(); prof = ops.PROFILE.summary(); ops.PROFILE.enabled = False
    tot = sum(v["ms"] for v in prof.values())
    print(f"{name}: {n} frames {prec}: {ms:.3f} ms  ({n / ms * 1e3:.0f} frames/s, {fl * n / ms / 1e9:.0f} TFLOP/s of the reference's FLOPs); bracketed GEMM launches {tot:.3f} ms")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        extra = f"{v['flops'] / v['ms'] / 1e9:8.1f} TFLOP/s  {v['bytes'] / v['ms'] / 1e9:6.2f} TB/s" if v["flops"] else ""
        print(f"   {k:60s} {v['calls']:3d} calls {v['ms']:8.3f} ms {extra}")
