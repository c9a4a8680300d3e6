"""Timestep respacing (reference improved_diffusion/respace.py): integer step sets must be bit-exact."""
import numpy as np
import torch as th

from ._lib import check, lib, ptr, stream
from .gaussian_diffusion import GaussianDiffusion


def _ddim_stride_set(num_timesteps, count):
    """'ddimN': the first integer stride whose range has exactly N entries (reference respace.py:30-37)."""
    for stride in range(1, num_timesteps):
        kept = range(0, num_timesteps, stride)
        if len(kept) == count:
            return set(kept)
    raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")


def space_timesteps(num_timesteps, section_counts):
    """Set of retained steps of the original process (reference respace.py:7-61).

    A string "ddimN" uses the DDIM paper's fixed stride; otherwise the T steps are cut into len(counts)
    sections and each contributes `count` steps at a fractional stride that is accumulated in a Python
    float and rounded with round() (banker's rounding) — reproduced operation for operation because the
    resulting integer sets must match exactly ("250" != "ddim250")."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            return _ddim_stride_set(num_timesteps, int(section_counts[len("ddim"):]))
        section_counts = [int(x) for x in section_counts.split(",")]
    n_sec = len(section_counts)
    base, extra = divmod(num_timesteps, n_sec)
    kept, start = [], 0
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process that skips steps of a base process (reference respace.py:64-110)."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base_ac = np.cumprod(1.0 - np.array(kwargs["betas"], dtype=np.float64), axis=0)
        self.timestep_map = [i for i in range(self.original_num_steps) if i in self.use_timesteps]
        prev = np.concatenate([[1.0], base_ac[self.timestep_map][:-1]])
        new_betas = [1 - a / p for a, p in zip(base_ac[self.timestep_map], prev)]       # beta'_i = 1 - abar_i / abar_prev_kept
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def _model_eps(self, model, *args, **kwargs):
        return super()._model_eps(self._wrap_model(model), *args, **kwargs)

    def training_losses(self, model, *args, **kwargs):
        return super().training_losses(self._wrap_model(model), *args, **kwargs)

    def _vb_terms_bpd(self, model, *args, **kwargs):           # reached through p_mean_variance in the reference (respace.py:90-93)
        return super()._vb_terms_bpd(self._wrap_model(model), *args, **kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def _scale_timesteps(self, t):
        return t          # scaling is done by the wrapped model


_MAP_CACHE = {}


class _WrappedModel:
    """Maps spaced step indices to original ones before calling the network (reference respace.py:112-124).
    The map lives on the device (the reference rebuilds and uploads it on every call)."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model, self.timestep_map = model, timestep_map
        self.rescale_timesteps, self.original_num_steps = rescale_timesteps, original_num_steps

    def parameters(self):
        return self.model.parameters()

    def _map(self, device):
        key = (id(self.timestep_map), str(device))
        m = _MAP_CACHE.get(key)
        if m is None or m[0] is not self.timestep_map:
            m = (self.timestep_map, th.tensor(self.timestep_map, dtype=th.int64, device=device))
            _MAP_CACHE[key] = m
        return m[1]

    def map_timesteps(self, ts):
        """-> (int64 original indices, what the network sees: float * 1000/T_orig if rescale else the indices)."""
        ts = ts.to(th.int64).contiguous()
        n = ts.shape[0]
        out_f = th.empty(n, dtype=th.float32, device=ts.device)
        out_i = th.empty(n, dtype=th.int64, device=ts.device)
        check(lib.cdae_model_timesteps(ptr(ts), ptr(self._map(ts.device)), 1000.0 / self.original_num_steps,
                                       1 if self.rescale_timesteps else 0, ptr(out_f), ptr(out_i), n, stream()))
        return out_i, (out_f if self.rescale_timesteps else out_i)

    def __call__(self, x, ts, **kwargs):
        return self.model(x, self.map_timesteps(ts)[1], **kwargs)
