"""Drop-in alias: `improved_diffusion.fp16_util` -> causaldiffae_amd.fp16_util (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import fp16_util as _impl

sys.modules[__name__] = _impl
