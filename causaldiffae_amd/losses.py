"""Likelihood helpers with the reference's names (improved_diffusion/losses.py:12-77).  The training / evaluation path does
NOT go through these: the bound is one fused kernel (`cdae_vb_terms`, csrc/elementwise.hip).  They exist for callers that
import them directly (`_prior_bpd`, user scripts) and run as plain device tensor ops."""
import numpy as np
import torch as th


def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL(N(mean1, e^logvar1) || N(mean2, e^logvar2)), broadcasting; at least one argument must be a tensor."""
    tensor = next((o for o in (mean1, logvar1, mean2, logvar2) if isinstance(o, th.Tensor)), None)
    assert tensor is not None, "at least one argument must be a Tensor"
    logvar1, logvar2 = [v if isinstance(v, th.Tensor) else th.tensor(v).to(tensor) for v in (logvar1, logvar2)]
    return 0.5 * (-1.0 + logvar2 - logvar1 + th.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * th.exp(-logvar2))


def approx_standard_normal_cdf(x):
    """tanh approximation of the standard normal CDF."""
    return 0.5 * (1.0 + th.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * th.pow(x, 3))))


def discretized_gaussian_log_likelihood(x, *, means, log_scales):
    """log P(x) of a Gaussian discretised to 8-bit bins on [-1, 1] (bin half-width 1/255, open-ended edge bins)."""
    assert x.shape == means.shape == log_scales.shape
    centered = x - means
    inv_stdv = th.exp(-log_scales)
    cdf_plus = approx_standard_normal_cdf(inv_stdv * (centered + 1.0 / 255.0))
    cdf_min = approx_standard_normal_cdf(inv_stdv * (centered - 1.0 / 255.0))
    inner = th.log((cdf_plus - cdf_min).clamp(min=1e-12))
    upper = th.log((1.0 - cdf_min).clamp(min=1e-12))
    lower = th.log(cdf_plus.clamp(min=1e-12))
    return th.where(x < -0.999, lower, th.where(x > 0.999, upper, inner))
