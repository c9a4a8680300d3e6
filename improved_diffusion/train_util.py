"""Drop-in alias: `improved_diffusion.train_util` -> causaldiffae_amd.train_util (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import train_util as _impl

sys.modules[__name__] = _impl
