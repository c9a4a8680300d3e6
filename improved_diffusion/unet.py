"""Drop-in alias: `improved_diffusion.unet` -> causaldiffae_amd.unet (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import unet as _impl

sys.modules[__name__] = _impl
