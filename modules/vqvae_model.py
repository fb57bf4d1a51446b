from mage_amd.modules.vqvae_model import *  # noqa: F401,F403
from mage_amd.modules.vqvae_model import VectorQuantizedVAE  # noqa: F401
