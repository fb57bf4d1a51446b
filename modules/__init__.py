"""Drop-in shim: `modules.mage_model` / `modules.vqvae_model` resolve to the MI355X-native classes
(so the reference's yaml `target:` strings and `from modules.mage_model import MAGE` keep working)."""
