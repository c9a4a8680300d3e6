"""Data feed.  BASELINE benchmarks use synthetic tensors of the named shapes (no datasets ship with the repo or
the reference); dataset readers (MorphoMNIST idx, Pendulum PNG, CausalCircuit npz) are SURVEY §8f.4 "next"."""
import numpy as np
import torch as th


def load_data(*, data_dir, batch_size, image_size, class_cond=False, deterministic=False, in_channels=3, n_vars=4, seed=0):
    """Infinite generator of (batch [N,C,S,S] in [0,1], cond dict) like the reference's load_data
    (image_datasets.py:69-126).  data_dir "" / "synthetic" -> seeded synthetic images + labels."""
    if data_dir not in ("", "synthetic", None):
        raise NotImplementedError("dataset readers are the next widening step (SURVEY §8f.4); use data_dir='synthetic'")
    rng = np.random.RandomState(seed)
    while True:
        x = th.from_numpy(rng.rand(batch_size, in_channels, image_size, image_size).astype(np.float32))
        cond = {"c": th.from_numpy(rng.rand(batch_size, n_vars).astype(np.float32))}
        if class_cond:
            cond["y"] = th.from_numpy(rng.randint(0, 10, size=(batch_size,)).astype(np.int64))
        yield x, cond
