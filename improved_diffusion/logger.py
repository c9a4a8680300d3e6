"""Drop-in alias: `improved_diffusion.logger` -> causaldiffae_amd.logger (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import logger as _impl

sys.modules[__name__] = _impl
