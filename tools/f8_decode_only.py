"""f8 VQ-VAE decode only (992 frames, bf16), a few calls: the command the rocprofv3 / PMC scripts wrap.  Tuning only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd.modules.vqvae_model import VectorQuantizedVAE
from mage_amd.utils import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 992
dev = torch.device("cuda", 0)
vq = VectorQuantizedVAE(3, 8, 256, 512).eval()
synth.fill_state_dict(vq, 0)
vq = vq.to(dev).set_precision("bf16")
ids = torch.randint(0, 512, (N, 16, 16), device=dev)
for _ in range(3): vq.decode(ids)
torch.cuda.synchronize()
