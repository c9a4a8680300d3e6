"""Drop-in alias: `improved_diffusion.metrics` -> causaldiffae_amd.metrics (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import metrics as _impl

sys.modules[__name__] = _impl
