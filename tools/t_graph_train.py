"""Dev: eager vs hipGraph training steps from the same RNG state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from improved_diffusion import script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop
dev = torch.device("cuda:0")
def make(use_graph):
    cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True}
    model, diff = su.create_model_and_diffusion(**cfg)
    bench.randomize(model, 4321)
    model.to(dev).train()
    data = load_data(data_dir="synthetic", batch_size=8, image_size=64, in_channels=3, n_vars=4, seed=0)
    loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=8, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                     save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3, use_graph=use_graph)
    diff.kl_weight = 0.1
    return loop, data
res = []
for ug in (False, False, True):
    loop, data = make(ug)
    losses, grads = [], []
    for i in range(6):
        np.random.seed(100 + i); torch.manual_seed(200 + i)
        b, c = next(data)
        loop.forward_backward(b, c)
        grads.append(loop.opt.flat.grad.clone())
        loop.optimize_normal()
        losses.append(float(loop.last_losses["loss"].mean().item()))
    print("graph" if ug else "eager", "graphs:", len(loop._graphs), "failed:", loop._graph_failed, losses)
    res.append((losses, loop.opt.flat.flat.clone(), grads))
for a, b, name in ((0, 1, "eager vs eager"), (0, 2, "eager vs graph")):
    print(name, "max |param diff|", (res[a][1] - res[b][1]).abs().max().item(),
          "grad rel diff per step", [float((x - y).abs().max() / x.abs().max()) for x, y in zip(res[a][2], res[b][2])])
