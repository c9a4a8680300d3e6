"""Dev tool: a few eager DDIM steps of the P64 bench workload (batch 128) for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import causaldiffae_amd  # noqa: F401
from improved_diffusion import script_util as su
import bench

cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 4, "n_vars": 4, "rep_cond": True,
       "causal_modeling": True, "timestep_respacing": "ddim100"}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 1234)
model.to("cuda:0").eval()
N = int(os.environ.get("B", "128"))
x = torch.randn(N, 4, 64, 64, device="cuda:0")
z = torch.randn(N, 512, device="cuda:0")
t = torch.full((N,), 50, dtype=torch.int64, device="cuda:0")
with torch.no_grad():
    for _ in range(int(os.environ.get("STEPS", "3"))):
        diff.ddim_sample(model, x, t, model_kwargs={"z": z})
torch.cuda.synchronize()
