"""Stand-in latent first stage for the MAGE+ (use_cids=False) tests.

MAGE+ sits on an `ldm` AutoencoderKL (config/mage+_caterv2.yaml:23-45) whose source is not in the reference mount
(requirements.txt:21, unpinned): that component is outside the build and its parity is unpinned.  The MAGE side
only needs `embed_dim`, `encode(x) -> latents [N, embed_dim, h, w]` and `decode(z) -> images`; this fixed, parameter-free
module provides them with plain torch ops so that the SAME object can sit under the reference's MAGE (golden generation)
and under the MI355X MAGE (GPU tests).  It is test infrastructure, not product code.
"""
import torch
import torch.nn.functional as F
from torch import nn


class StandInLatentFirstStage(nn.Module):
    def __init__(self, embed_dim: int = 4, down: int = 8, in_ch: int = 3):
        super().__init__()
        self.embed_dim, self.down, self.in_ch = embed_dim, down, in_ch
        g = torch.Generator().manual_seed(1234)
        self.register_buffer("mix", torch.randn(embed_dim, in_ch, generator=g) * 0.8, persistent=False)

    def encode(self, x):
        p = F.avg_pool2d(x.float(), self.down)
        return torch.einsum("ec,nchw->nehw", self.mix.to(x.device), p)

    def decode(self, z):
        img = torch.einsum("ec,nehw->nchw", self.mix.to(z.device), z.float())
        return torch.tanh(F.interpolate(img, scale_factor=self.down, mode="nearest"))
