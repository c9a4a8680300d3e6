"""causaldiffae_amd — MI355X-native (gfx950) implementation of the CausalDiffAE diffusion hot path.

Importing the package loads causaldiffae_amd/libcdae.so (hand-written HIP kernels, C-ABI in include/cdae.h);
there is no eager / CPU fallback.  `improved_diffusion` at the repository root re-exports these modules
under the reference's package name."""
from . import _lib  # noqa: F401  (fails loudly if the HIP library is missing)
from ._lib import CdaeError, CdaeRangeError, get_precision, range_check, set_precision  # noqa: E402,F401
