"""f8 VQ-VAE encode only (248 frames of 128x128), a few calls: the command the rocprofv3 / PMC scripts wrap.  Tuning only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mage_amd.modules.vqvae_model import VectorQuantizedVAE
from mage_amd.utils import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 248
dev = torch.device("cuda", 0)
vq = VectorQuantizedVAE(3, 8, 256, 512).eval()
synth.fill_state_dict(vq, 0)
vq = vq.to(dev)
x = torch.rand(N, 3, 128, 128, device=dev) * 2 - 1
for _ in range(3): vq.encode(x)
torch.cuda.synchronize()
