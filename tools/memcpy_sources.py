"""Dev: who issues the hipMemcpy / hipMemset calls of a C64 batch-32 training step (torch.profiler, runtime events folded by the
enclosing op and the innermost stack frames inside this repo)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from improved_diffusion import script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop
dev = torch.device("cuda:0")
cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 4321)
model.to(dev).train()
data = load_data(data_dir="synthetic", batch_size=32, image_size=64, in_channels=3, n_vars=4, seed=0, device=dev)
loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=32, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                 save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3)
diff.kl_weight = 0.1
def steps(n):
    for _ in range(n):
        b, c = next(data); loop.forward_backward(b, c); loop.optimize_normal()
    torch.cuda.synchronize()
steps(3)
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    steps(N)
names = collections.Counter()
who = collections.Counter()
for e in prof.events():
    if not (e.name.startswith("hipMemcpy") or e.name.startswith("hipMemset") or "Memcpy" in e.name or "Memset" in e.name):
        continue
    names[e.name] += 1
    if not e.name.startswith("hip"):
        continue
    p, chain, stack = e.cpu_parent, [], None
    while p is not None:
        chain.append(p.name)
        if stack is None and p.stack:
            stack = [s for s in p.stack if "/repo/" in s or "causaldiffae" in s or "improved_diffusion" in s][:3]
        p = p.cpu_parent
    who[(e.name, " < ".join(chain[:3]), " | ".join(stack or []))] += 1
print({k: v / N for k, v in names.items()})
for (n, chain, st), c in who.most_common(40):
    print(f"{c / N:6.1f}  {n}  [{chain}]  {st}")
