"""Drop-in alias: `improved_diffusion.dist_util` -> causaldiffae_amd.dist_util (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import dist_util as _impl

sys.modules[__name__] = _impl
