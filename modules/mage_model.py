from mage_amd.modules.mage_model import *  # noqa: F401,F403
from mage_amd.modules.mage_model import MAGE, FlatAxialDecoder, MAEncoder, TransformerTextEncoder  # noqa: F401
