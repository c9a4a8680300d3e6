"""Dev: a few hundred training steps on synthetic data — the loss must fall and stay finite (sanity check of the training kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from improved_diffusion import script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True}
model, diff = su.create_model_and_diffusion(**cfg)
model.to(dev).train()                      # the factory's own initialisation (zero-init output convs), like a real run
data = load_data(data_dir="synthetic", batch_size=32, image_size=64, in_channels=3, n_vars=4, seed=0, device=dev)
loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=32, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                 save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3,
                 use_fp16=os.environ.get("FP16") == "1")          # FP16=1: on the 16-bit torso
hist = []
t0 = time.perf_counter()
for i in range(steps):
    b, c = next(data)
    loop.run_step(b, c)
    loop.step += 1
    diff.kl_weight = min(1.0, loop.step / 49999)
    if i % 25 == 0 or i == steps - 1:
        l = loop.last_losses
        hist.append((i, float(l["mse"].mean()), float(l["kld_rep"].mean())))
        rss = int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") >> 20      # host memory: the runtime's command batches must not pile up
        print(hist[-1], "host RSS MiB", rss, "device MiB", torch.cuda.memory_allocated() >> 20, flush=True)
torch.cuda.synchronize()
print("steps/s", steps / (time.perf_counter() - t0))
assert all(map(lambda h: h[1] == h[1] and h[1] < 10, hist)), "non-finite or exploding loss"
assert hist[-1][1] < 0.6 * hist[0][1], "mse did not fall"
print("ok: mse", hist[0][1], "->", hist[-1][1])
