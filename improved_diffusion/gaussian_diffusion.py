"""Drop-in alias: `improved_diffusion.gaussian_diffusion` -> causaldiffae_amd.gaussian_diffusion (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import gaussian_diffusion as _impl

sys.modules[__name__] = _impl
