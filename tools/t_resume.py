"""Dev: checkpoint save -> resume round trip through TrainLoop (model + EMA by parameter NAME; the flat buffer's internal order is free)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from improved_diffusion import logger, script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop
dev = torch.device("cuda:0")
d = tempfile.mkdtemp()
logger.configure(dir=d)
cfg = {**su.model_and_diffusion_defaults(), "image_size": 32, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True, "num_channels": 64}
def make(resume):
    model, diff = su.create_model_and_diffusion(**cfg)
    if not resume:
        bench.randomize(model, 4321)
    model.to(dev).train()
    data = load_data(data_dir="synthetic", batch_size=4, image_size=32, in_channels=3, n_vars=4, seed=0, device=dev)
    return TrainLoop(model=model, diffusion=diff, data=data, batch_size=4, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                     save_interval=10 ** 9, resume_checkpoint=resume, rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3), data
loop, data = make("")
for _ in range(3):
    b, c = next(data); loop.run_step(b, c); loop.step += 1
loop.save()
path = os.path.join(d, f"model{loop.step:06d}.pt")
assert os.path.exists(path), os.listdir(d)
sd = {k: v.detach().clone() for k, v in loop.model.state_dict().items()}
ema = {k: v.detach().clone() for k, v in loop.opt.ema_state_dict(0).items()}
loop2, data2 = make(path)
assert loop2.resume_step == loop.step
for k, v in loop2.model.state_dict().items():
    assert torch.equal(v.cpu(), sd[k].cpu()), k
for k, v in loop2.opt.ema_state_dict(0).items():
    if v.dtype.is_floating_point and "running" not in k and "num_batches" not in k:
        assert torch.equal(v.cpu(), ema[k].cpu()), k
b, c = next(data2); loop2.run_step(b, c)
print("resume ok: step", loop2.resume_step, "loss", float(loop2.last_losses["loss"].mean()))
