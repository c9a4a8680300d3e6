"""Drop-in alias: `improved_diffusion.script_util` -> causaldiffae_amd.script_util (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import script_util as _impl

sys.modules[__name__] = _impl
