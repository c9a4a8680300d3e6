"""Drop-in alias: `improved_diffusion.resample` -> causaldiffae_amd.resample (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import resample as _impl

sys.modules[__name__] = _impl
