"""Drop-in alias: `improved_diffusion.image_datasets` -> causaldiffae_amd.image_datasets (reference module name kept so the reference's
scripts import unchanged)."""
import sys

from causaldiffae_amd import image_datasets as _impl

sys.modules[__name__] = _impl
