"""Alias: improved_diffusion.counterfactual -> causaldiffae_amd.counterfactual."""
import sys

from causaldiffae_amd import counterfactual as _impl

sys.modules[__name__] = _impl
