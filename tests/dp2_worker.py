"""Worker of tests/test_gpu_model.py::test_data_parallel_gradients_match_single_process: 2-rank data-parallel step (gloo over one GPU, both ranks the same batch) vs one process: the all-reduced, averaged flat gradient must match."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist

BACKEND = os.environ.get("DP2_BACKEND", "gloo")
SPLIT = os.environ.get("DP2_SPLIT", "0") == "1"           # every rank its OWN 4 images (else the same 4)
# nccl (= RCCL) needs one GPU per rank; the gloo variant runs both ranks on cuda:0
DEV = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}" if BACKEND == "nccl" else "cuda:0"


def build(batch):
    import bench
    from improved_diffusion import script_util as su
    from improved_diffusion.train_util import TrainLoop
    cfg = {**su.model_and_diffusion_defaults(), "image_size": 32, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True,
           "num_channels": 64}
    model, diff = su.create_model_and_diffusion(**cfg)
    bench.randomize(model, 4321)
    model.to(DEV).train()
    loop = TrainLoop(model=model, diffusion=diff, data=iter(()), batch_size=batch, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                     save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3, bucket_mb=4)
    diff.kl_weight = 0.1
    return model, diff, loop

def data():
    g = torch.Generator().manual_seed(7)
    return torch.rand(8, 3, 32, 32, generator=g) * 2 - 1, {"c": torch.rand(8, 4, generator=g)}, torch.randint(0, 1000, (8,), generator=g), torch.randn(8, 3, 32, 32, generator=g)

def step(loop, diff, model, x, c, t, noise):
    dev = DEV
    loop.opt.zero_grad()
    loop.buckets.enabled = True
    torch.manual_seed(5)                         # encoder reparameterisation noise: same stream on every process (per-sample independent below)
    losses = diff.training_losses(model, x.to(dev), t.to(dev), model_kwargs={k: v.to(dev) for k, v in c.items()}, noise=noise.to(dev), rep_cond=True,
                                  causal_modeling=True)
    losses["loss"].mean().backward()
    loop.buckets.finish()
    return loop.opt.flat.grad.clone(), loop.opt.flat.names

if __name__ == "__main__":
    world = int(os.environ.get("WORLD_SIZE", "1"))
    x, c, t, noise = data()
    shard = lambda r: slice(4 * r, 4 * r + 4) if SPLIT else slice(0, 4)
    if world > 1:
        if BACKEND == "nccl":
            torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        dist.init_process_group(BACKEND, init_method="env://")
        r = dist.get_rank()
        model, diff, loop = build(4)
        sl = shard(r)          # SPLIT = 0: the SAME 4 images on every rank (the encoder's BatchNorm uses batch statistics): mean == single-process
        g, names = step(loop, diff, model, x[sl], {k: v[sl] for k, v in c.items()}, t[sl], noise[sl])
        if r == 0:
            torch.save((g.cpu(), names, dist.get_backend()), os.environ["DP2_OUT"])
        dist.barrier()
    else:
        model, diff, loop = build(4)
        # the data-parallel result = mean over ranks of the per-shard gradients (each shard with its own BatchNorm batch statistics)
        acc = None
        for r in range(2 if SPLIT else 1):
            sl = shard(r)
            g, names = step(loop, diff, model, x[sl], {k: v[sl] for k, v in c.items()}, t[sl], noise[sl])
            acc = g.clone() if acc is None else acc + g
        g = (acc / (2 if SPLIT else 1)).cpu()
        g2, names2, backend = torch.load(os.environ["DP2_OUT"])
        assert names == names2 and backend == BACKEND, (backend, BACKEND)
        offs = loop.opt.flat.offsets + [loop.opt.flat.numel]
        worst = []
        for i, n in enumerate(names):
            a, b = g[offs[i]:offs[i + 1]], g2[offs[i]:offs[i + 1]]
            if "rep_emb.encoder" in n and n.endswith(".0.bias"):
                continue                      # conv bias in front of a BatchNorm: its gradient is exactly zero in theory, rounding noise in practice
            worst.append(((a - b).abs().max().item() / (a.abs().max().item() + 1e-30), n))
        worst.sort()
        print("params", len(names), "backend", backend, "worst rel diff", worst[-3:])
        assert worst[-1][0] < 1e-4, worst[-3:]
