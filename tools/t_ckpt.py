"""Dev: use_checkpoint=True (activation recomputation) with the fused training nodes — same gradients as without?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from improved_diffusion import script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop
dev = torch.device("cuda:0")
def run(ck):
    cfg = {**su.model_and_diffusion_defaults(), "image_size": 32, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True,
           "num_channels": 64, "use_checkpoint": ck}
    model, diff = su.create_model_and_diffusion(**cfg)
    bench.randomize(model, 4321)
    model.to(dev).train()
    data = load_data(data_dir="synthetic", batch_size=4, image_size=32, in_channels=3, n_vars=4, seed=0)
    loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=4, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                     save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3)
    diff.kl_weight = 0.1
    np.random.seed(100); torch.manual_seed(200)
    b, c = next(data)
    loop.forward_backward(b, c)
    return float(loop.last_losses["loss"].mean()), {n: p.grad.detach().clone() for n, p in model.named_parameters()}
(l0, g0), (l1, g1) = run(False), run(True)
rel = sorted(((g0[n] - g1[n]).abs().max().item() / (g0[n].abs().max().item() + 1e-30), n) for n in g0 if not ("rep_emb.encoder" in n and n.endswith(".0.bias")))
print("loss", l0, l1, "worst", rel[-3:])
