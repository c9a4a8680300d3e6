"""Drop-in alias: `improved_diffusion.losses` -> causaldiffae_amd.losses (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import losses as _impl

sys.modules[__name__] = _impl
