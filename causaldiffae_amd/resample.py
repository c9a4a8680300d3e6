"""Timestep samplers (reference improved_diffusion/resample.py).  The uniform sampler is the one on the hot path; the
loss-second-moment importance sampler (resample.py:71-154) is host-side bookkeeping over a [T, history] table."""
from abc import ABC, abstractmethod

import numpy as np
import torch as th


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    if name == "loss-second-moment":
        return LossSecondMomentResampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")


class _PinnedRing:
    """Host staging for the per-step timestep / weight uploads.  `from_numpy(x).to(cuda)` copies from pageable memory, which blocks
    the host until the stream has drained — every training step then starts with an empty launch queue.  A small ring of pinned
    buffers and `copy_(non_blocking=True)` keeps the upload asynchronous; a slot is reused only after its own copy has completed."""

    def __init__(self, depth=8):
        self.depth, self.slots, self.next = depth, {}, 0

    def upload(self, arr, device):
        key = (arr.dtype.str, arr.shape)
        ring = self.slots.get(key)
        if ring is None:
            ring = self.slots[key] = [[th.empty(arr.shape, dtype=th.from_numpy(arr[:0]).dtype).pin_memory(), None] for _ in range(self.depth)]
        self.next = (self.next + 1) % self.depth
        buf, ev = ring[self.next]
        if ev is not None:
            ev.synchronize()
        buf.numpy()[...] = arr
        out = th.empty(arr.shape, dtype=buf.dtype, device=device)
        out.copy_(buf, non_blocking=True)
        ev = th.cuda.Event()
        ev.record()
        ring[self.next][1] = ev
        return out


_RING = _PinnedRing()


def _upload(arr, device):
    """numpy -> device tensor; asynchronous on CUDA devices (see _PinnedRing), plain conversion elsewhere."""
    if th.device(device).type != "cuda":
        return th.from_numpy(np.ascontiguousarray(arr)).to(device)
    return _RING.upload(np.ascontiguousarray(arr), device)


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
        return _upload(idx.astype(np.int64, copy=False), device), _upload(weights.astype(np.float32), device)


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


class LossAwareSampler(ScheduleSampler):
    def update_with_local_losses(self, local_ts, local_losses):
        """Every rank contributes its (timesteps, losses); all ranks then apply the same update in rank order so the
        reweighting stays identical everywhere (resample.py:72-103, done with one object all-gather instead of three padded
        tensor all-gathers)."""
        import torch.distributed as dist
        mine = (local_ts.detach().cpu().tolist(), local_losses.detach().cpu().tolist())
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            parts = [None] * dist.get_world_size()
            dist.all_gather_object(parts, mine)
        else:
            parts = [mine]
        self.update_with_all_losses([t for p in parts for t in p[0]], [l for p in parts for l in p[1]])

    @abstractmethod
    def update_with_all_losses(self, ts, losses):
        """Deterministic update from the gathered (timestep, loss) pairs; called with identical arguments on every rank."""


class LossSecondMomentResampler(LossAwareSampler):
    """p(t) proportional to sqrt(E[loss_t^2]) over the last `history_per_term` losses seen at t, mixed with a uniform floor;
    uniform until every timestep has a full history (resample.py:127-154).  The history is a ring per timestep — the
    second moment does not depend on the order the reference keeps by shifting.  (The reference's constructor uses
    `np.int`, removed in numpy >= 1.24, so it cannot be instantiated in this image: parity is by definition.)"""

    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion, self.history_per_term, self.uniform_prob = diffusion, history_per_term, uniform_prob
        self._loss_history = np.zeros([diffusion.num_timesteps, history_per_term], dtype=np.float64)
        self._loss_counts = np.zeros([diffusion.num_timesteps], dtype=np.int64)       # total losses ever seen per step

    def weights(self):
        if not self._warmed_up():
            return np.ones([self.diffusion.num_timesteps], dtype=np.float64)
        w = np.sqrt(np.mean(self._loss_history ** 2, axis=-1))
        w /= np.sum(w)
        w *= 1 - self.uniform_prob
        w += self.uniform_prob / len(w)
        return w

    def update_with_all_losses(self, ts, losses):
        for t, loss in zip(ts, losses):
            self._loss_history[t, self._loss_counts[t] % self.history_per_term] = loss
            self._loss_counts[t] += 1

    def _warmed_up(self):
        return bool((self._loss_counts >= self.history_per_term).all())
