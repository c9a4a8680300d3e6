"""Timestep samplers (reference improved_diffusion/resample.py).  Only the uniform sampler is on the hot path."""
from abc import ABC, abstractmethod

import numpy as np
import torch as th


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    if name == "loss-second-moment":
        raise NotImplementedError("loss-second-moment resampling is out of scope (reference resample.py:134 uses removed np.int)")
    raise NotImplementedError(f"unknown schedule sampler: {name}")


class ScheduleSampler(ABC):
    @abstractmethod
    def weights(self):
        """Positive (unnormalised) weight per diffusion step."""

    def sample(self, batch_size, device):
        """Importance-sample timesteps with numpy's global RNG like the reference (resample.py:44-60):
        -> (int64 indices [B], float32 weights [B]) on `device`."""
        w = self.weights()
        p = w / np.sum(w)
        idx = np.random.choice(len(p), size=(batch_size,), p=p)
        weights = 1 / (len(p) * p[idx])
        return th.from_numpy(idx).long().to(device), th.from_numpy(weights).float().to(device)


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


class LossAwareSampler(ScheduleSampler):
    """Kept as a type so `isinstance(sampler, LossAwareSampler)` in callers keeps working."""

    def update_with_local_losses(self, local_ts, local_losses):
        raise NotImplementedError
