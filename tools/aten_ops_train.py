"""Dev: which ATen ops (not libcdae launches) a C64 batch-32 training step issues, with counts and host time (torch.profiler)."""
import os, sys
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
N = 3
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    steps(N)
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
print(f"{'op':60s} {'calls/step':>10s} {'self us/step':>12s} {'total us/step':>13s}")
for e in rows[:45]:
    print(f"{e.key[:60]:60s} {e.count / N:10.1f} {e.self_cpu_time_total / N:12.1f} {e.cpu_time_total / N:13.1f}")
