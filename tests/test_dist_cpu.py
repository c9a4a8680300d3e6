"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: flat parameter/gradient buffers, bucketed
asynchronous gradient all-reduce launched from post-accumulate hooks, no_sync microbatch semantics, rank-0
broadcast, and batch sharding for sampling.  The model here is a tiny torch-CPU module: the product UNet cannot
run on CPU (no fallback), the reducer is model-agnostic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from causaldiffae_amd import dist_util
    from causaldiffae_amd.train_util import FlatParams, GradBuckets
    dist_util.setup_dist(backend="gloo")
    try:
        torch.manual_seed(100 + rank)                                  # different init per rank on purpose
        model = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                                    torch.nn.Linear(64, 8))
        conv = torch.nn.Conv2d(4, 8, 3).to(memory_format=torch.channels_last)      # channels_last param keeps its strides
        model.add_module("conv", conv)
        flat = FlatParams(model)
        assert all(p.data_ptr() == flat.flat.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
        assert model.conv.weight.permute(0, 2, 3, 1).is_contiguous()
        dist.broadcast(flat.flat, 0)                                   # what FusedAdamWEMA.broadcast_from_rank0 does
        ref = [torch.zeros_like(flat.flat) for _ in range(world)]
        dist.all_gather(ref, flat.flat)
        assert torch.equal(ref[0], ref[1])
        buckets = GradBuckets(flat, bucket_bytes=1024)                 # several small buckets
        assert len(buckets.buckets) >= 3
        covered = sorted((b["lo"], b["hi"]) for b in buckets.buckets)
        assert covered[0][0] == 0 and covered[-1][1] == flat.numel and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))

        def loss_on(x):
            return model[:5](x).pow(2).mean() + model.conv(torch.ones(1, 4, 5, 5)).mean() * x.mean()

        g = torch.Generator().manual_seed(7)
        xs = [torch.randn(6, 16, generator=g) for _ in range(world)]          # same on every rank; rank r uses xs[r]
        # single-process reference: mean over ranks of each rank's gradient
        flat.zero_grad()
        buckets.enabled = False
        want = torch.zeros_like(flat.grad)
        for r in range(world):
            flat.zero_grad()
            loss_on(xs[r]).backward()
            want += flat.grad / world
        # distributed: two microbatches, all-reduce only with the last one (no_sync semantics)
        flat.zero_grad()
        buckets.reset()
        half = xs[rank][:3], xs[rank][3:]
        want_mb = torch.zeros_like(flat.grad)
        for r in range(world):
            flat.zero_grad()
            (loss_on(xs[r][:3]) + loss_on(xs[r][3:])).backward()
            want_mb += flat.grad / world
        flat.zero_grad()
        buckets.reset()
        buckets.enabled = False
        loss_on(half[0]).backward()
        buckets.enabled = True
        loss_on(half[1]).backward()
        assert all(b["work"] is not None for b in buckets.buckets)     # every bucket was launched from a hook during backward
        buckets.finish()
        err_mb = (flat.grad - want_mb).abs().max().item()
        # plain one-shot step
        flat.zero_grad()
        loss_on(xs[rank]).backward()
        buckets.finish()
        err = (flat.grad - want).abs().max().item()
        # overlap: while backward is still running (the gradient has only just reached the FIRST layer's output) the buckets of the later
        # layers must already have been handed to the process group — not queued up for finish()
        flat.zero_grad()
        seen = {}
        h0 = model[0](xs[rank])
        h0.register_hook(lambda g_: seen.update(launched=buckets.next_launch, in_flight=sum(b["work"] is not None for b in buckets.buckets),
                                                first_fired=buckets.fired[0]))
        (model[1:5](h0).pow(2).mean() + model.conv(torch.ones(1, 4, 5, 5)).mean() * xs[rank].mean()).backward()
        assert seen["launched"] >= 1 and seen["in_flight"] >= 1 and not seen["first_fired"], seen
        buckets.finish(average=False)                                   # TrainLoop's call: the SUM stays, the optimizer kernel scales by 1 / world
        err_sum = (flat.grad / world - want).abs().max().item()
        assert err_sum < 1e-6, err_sum
        lo, hi = dist_util.shard_range(1024 + 3, rank, world)
        got = dist_util.gather_samples(torch.full((2, 3), float(rank)))
        # uneven shards (n % world != 0: the low rank owns one row more) — every rank must still see all rows in rank order
        l5, h5 = dist_util.shard_range(5, rank, world)
        uneven = dist_util.gather_samples(torch.arange(l5, h5, dtype=torch.float32).reshape(-1, 1).repeat(1, 4))
        assert [t.shape[0] for t in uneven] == [3, 2] and torch.equal(torch.cat(uneven)[:, 0], torch.arange(5.0))
        q.put((rank, err, err_mb, (lo, hi), [float(t[0, 0]) for t in got]))
    except Exception as e:          # surface the failure instead of a queue timeout
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for r in res:
        assert r[1] != "error", r[2]
    res = sorted(res)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, err_mb, rng, gathered in res:
        assert err < 1e-6 and err_mb < 1e-6, (rank, err, err_mb)
        assert gathered == [0.0, 1.0]
    assert res[0][3] == (0, 514) and res[1][3] == (514, 1027)          # disjoint cover, remainder to the low rank


def test_shard_range_covers():
    from causaldiffae_amd.dist_util import shard_range
    for n in (1, 7, 128, 1024, 1027):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_kl_weight_and_resume_name(golden):
    import numpy as np
    from causaldiffae_amd.train_util import linear_kl_weight, parse_resume_step_from_filename
    g = golden("g7_train.npz")
    got = np.array([linear_kl_weight(s, 50000, 0.0, 1.0) for s in (0, 1, 2, 25000, 49999, 50000, 60000)])
    np.testing.assert_array_equal(got, g["kl_weight_sched"])
    assert parse_resume_step_from_filename("/x/y/model012345.pt") == 12345
    assert parse_resume_step_from_filename("/x/y/ema_checkpoint.pt") == 0


def _worker_feed(rank, world, port, root, logdir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from causaldiffae_amd import dist_util, logger
    from causaldiffae_amd.image_datasets import load_data, read_morphomnist
    from causaldiffae_amd.resample import LossSecondMomentResampler
    dist_util.setup_dist(backend="gloo")
    try:
        # (1) load_data takes the [rank::world] stride of the files
        feed = load_data(data_dir=root, batch_size=4, image_size=28, split="train")
        x, cond = next(feed)
        shard = read_morphomnist(root, "train", rank, world)
        full = read_morphomnist(root, "train")
        ok_shard = bool((shard.cond["c"] == full.cond["c"][rank::world]).all()) and x.shape == (4, 1, 28, 28)
        seen = {tuple(row.tolist()) for row in cond["c"]}
        mine = {tuple(row.tolist()) for row in torch.from_numpy(shard.cond["c"])}
        # (2) logger: means are averaged over ranks weighted by their counts, delivered on rank 0
        logger.configure(dir=logdir, format_strs=["csv"], comm=True)
        logger.logkv_mean("loss", 1.0 + rank)            # rank 0: one sample of 1.0; rank 1: two samples of 2.0 and 4.0 (mean 3.0)
        if rank == 1:
            logger.logkv_mean("loss", 5.0 - rank)
        kv = logger.dumpkvs()
        logger.reset()
        # (3) loss-aware sampler: both ranks end with the same history after exchanging their local losses
        class D:
            num_timesteps = 6
        s = LossSecondMomentResampler(D, history_per_term=2)
        for k in range(3):
            ts = torch.tensor([(rank + 2 * k) % 6, (rank + 2 * k + 3) % 6])
            s.update_with_local_losses(ts, torch.tensor([1.0 + rank + k, 2.0 * (1 + rank) + k]))
        hist = torch.from_numpy(s._loss_history.copy())
        hists = [torch.zeros_like(hist) for _ in range(world)]
        dist.all_gather(hists, hist)
        q.put((rank, ok_shard, seen <= mine, kv, bool(torch.equal(hists[0], hists[1])), int(s._loss_counts.sum())))
    except Exception:
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None, None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_feed_logger_and_sampler_world2(tmp_path):
    """Rank-strided dataset shards, cross-rank averaged logging and the loss-aware sampler's exchange under gloo, world size 2."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_datasets_cpu import _make_morpho
    root = str(tmp_path / "morphomnist")
    _make_morpho(root, n_train=21, n_test=8)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_feed, args=(r, world, port, root, str(tmp_path / "logs"), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for r in res:
        assert r[1] != "error", r[2]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = sorted(res)
    for rank, ok_shard, batch_in_shard, kv, same_hist, n_losses in res:
        assert ok_shard and batch_in_shard and same_hist
        assert n_losses == 12                               # 3 rounds x 2 ranks x 2 losses, applied identically on both ranks
    assert abs(res[0][3]["loss"] - (1.0 * 1 + 3.0 * 2) / 3) < 1e-12          # weighted by sample counts, on rank 0
    assert res[1][3] == {"dummy": 1}
