"""Dev: BASELINE config[1] shape — MorphoMNIST 32x32 (1 channel, 2 causal variables, class-conditional) training, batch 256 on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from improved_diffusion import script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = {**su.model_and_diffusion_defaults(), "image_size": 32, "in_channels": 1, "n_vars": 2, "rep_cond": True, "causal_modeling": True}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 4321)
model.to(dev).train()
data = load_data(data_dir="synthetic", batch_size=B, image_size=32, in_channels=1, n_vars=2, seed=0, device=dev)
loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=B, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                 save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=2, causal_modeling=True, in_channels=1)
diff.kl_weight = 0.1
for _ in range(3):
    b, c = next(data); loop.forward_backward(b, c); loop.optimize_normal()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n):
    b, c = next(data); loop.forward_backward(b, c); loop.optimize_normal()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"M32 batch {B}: {1e3 * dt:.2f} ms/step, {1 / dt:.2f} steps/s, {B / dt:.0f} images/s, params {sum(p.numel() for p in model.parameters()) / 1e6:.1f}M, loss {float(loop.last_losses['loss'].mean()):.3f}")
