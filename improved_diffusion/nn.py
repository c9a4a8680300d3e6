"""Drop-in alias: `improved_diffusion.nn` -> causaldiffae_amd.nn (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import nn as _impl

sys.modules[__name__] = _impl
