"""Distributed bootstrap (reference improved_diffusion/dist_util.py, which used mpi4py + gloo): one process per
GPU, torchrun-style env rendezvous, backend "nccl" (= RCCL over xGMI on ROCm) on GPUs and gloo on CPU."""
import io
import os

import torch as th
import torch.distributed as dist

GPUS_PER_NODE = 8


def setup_dist(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (single process if unset)."""
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if th.cuda.is_available() else "gloo"
    if th.cuda.is_available():
        th.cuda.set_device(int(os.environ.get("LOCAL_RANK", int(os.environ["RANK"]) % max(1, th.cuda.device_count()))))
    dist.init_process_group(backend=backend, init_method="env://")


def dev():
    if th.cuda.is_available():
        return th.device(f"cuda:{th.cuda.current_device()}")
    return th.device("cpu")


def load_state_dict(path, **kwargs):
    """Rank 0 reads the file, every rank receives the bytes (the reference MPI-broadcasts them, dist_util.py:54-64)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return th.load(path, **kwargs)
    obj = [open(path, "rb").read() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(obj, src=0)
    return th.load(io.BytesIO(obj[0]), **kwargs)


def sync_params(params):
    """Broadcast tensors from rank 0 (a no-op in the reference, dist_util.py:67-74: fixed)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for p in params:
        with th.no_grad():
            dist.broadcast(p, 0)


def shard_range(n, rank=None, world=None):
    """[lo, hi) slice of a global batch of n independent items owned by `rank` (sampling shards the image batch
    over ranks with no collective in the loop; remainders go to the lowest ranks)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_samples(sample):
    """all_gather of finished samples (the reference does this after the loop, image_causaldae_test.py:438-439).  The ranks' shards
    may differ by one row (shard_range gives the remainder to the lowest ranks): every rank pads to the longest shard, and the
    gathered per-rank row counts trim the result, so the collective always sees equal shapes."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [sample]
    world = dist.get_world_size()
    sample = sample.contiguous()
    counts = th.zeros(world, dtype=th.int64, device=sample.device)
    counts[dist.get_rank()] = sample.shape[0]
    dist.all_reduce(counts)
    counts = counts.tolist()
    longest = max(counts)
    if sample.shape[0] < longest:
        pad = th.zeros((longest - sample.shape[0],) + tuple(sample.shape[1:]), dtype=sample.dtype, device=sample.device)
        sample = th.cat([sample, pad], dim=0)
    out = [th.zeros_like(sample) for _ in range(world)]
    dist.all_gather(out, sample)
    return [o[:n] for o, n in zip(out, counts)]
