"""Dev tool: which autograd nodes / Python call sites own the ATen copy / add / fill kernels of one C64 training step.
Uses torch.profiler's CPU op tree: every aten::copy_ / add / add_ / fill_ / zero_ / clone event is attributed to its top-level parent op."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda:0")
import numpy as np
from improved_diffusion import script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop
cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 4321)
model.to(dev).train()
data = load_data(data_dir="synthetic", batch_size=32, image_size=64, in_channels=3, n_vars=4, seed=0, device=dev)
loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=32, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                 save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3)
diff.kl_weight = 0.1
for _ in range(3):
    b, c = next(data); loop.forward_backward(b, c); loop.optimize_normal()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    b, c = next(data); loop.forward_backward(b, c); loop.optimize_normal()
    torch.cuda.synchronize()
want = {"aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::clone", "aten::mul", "aten::sum", "aten::cat", "aten::contiguous"}
cnt = collections.Counter()
for e in prof.events():
    if e.name not in want:
        continue
    chain, p = [], e.cpu_parent
    while p is not None:
        chain.append(p.name)
        p = p.cpu_parent
    if any(n in want for n in chain):        # count outermost only
        continue
    st = [s for s in (e.stack or []) if "causaldiffae_amd" in s or "bench" in s]
    shapes = str(e.input_shapes)[:60]
    cnt[(e.name, " < ".join(chain[:3])[:110], st[0][-60:] if st else "", shapes)] += 1
for k, c in cnt.most_common(60):
    print(c, *k, sep=" | ")
