"""Drop-in alias: `improved_diffusion.respace` -> causaldiffae_amd.respace (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import respace as _impl

sys.modules[__name__] = _impl
