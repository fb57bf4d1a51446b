"""Per-kernel time of ONE autoregressive_generate call (HIP events around every launch).  usage: call_profile.py [full|incremental] [bf16|f16x3] [B=64]"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
from mage_amd import ops
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config
dev = "cuda:0"
m = instantiate_from_config(synth.mnist_model_config(frames_length=16)).eval()
synth.fill_state_dict(m, 0)
m = m.to(dev).set_precision(prec)
m.ar_mode = mode
m.use_graph = False
b = {k: v.to(dev) for k, v in synth.synth_batch_mnist(B, 16, seed=100).items()}
for _ in range(2): m.autoregressive_generate(b)
torch.cuda.synchronize()
ops.PROFILE.reset(True)
m.autoregressive_generate(b)
torch.cuda.synchronize()
res = ops.PROFILE.summary()
tot = sum(v["ms"] for k, v in res.items() if k != "decoder_step")
print(f"{mode} {prec} B={B}: {sum(v['calls'] for k, v in res.items() if k != 'decoder_step')} bracketed launches, {tot:.3f} ms in kernels")
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:60s} calls {v['calls']:4d}  {v['ms']:8.3f} ms  avg {1e3*v['ms']/v['calls']:8.1f} us")
