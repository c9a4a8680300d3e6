"""Worker of tests/test_gpu_model.py::test_rccl_data_parallel_path_with_one_rank: three C64 training steps through TrainLoop
(a) without a process group and (b) in an initialised `nccl` (= RCCL) group of ONE rank with CDAE_DDP_FORCE=1, i.e. with the gradient hooks,
the ordered 64 MiB bucket launches on RCCL's stream overlapping backward, the waits and the broadcast of the initial state all live.  The
one-rank reduction is the identity, so both runs must end in the same weights; prints a JSON record."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import bench
from improved_diffusion import script_util as su
from improved_diffusion.train_util import TrainLoop

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
mode = sys.argv[1]
if mode == "rccl":
    os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", CDAE_DDP_FORCE="1")
    os.environ.setdefault("MASTER_PORT", "29631")
    dist.init_process_group("nccl", init_method="env://")
N = 4
g = torch.Generator().manual_seed(21)
batches = [(torch.rand(N, 3, 64, 64, generator=g) * 2 - 1, {"c": torch.rand(N, 4, generator=g)}) for _ in range(3)]
cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 4321)
model.to(dev).train()
loop = TrainLoop(model=model, diffusion=diff, data=iter(()), batch_size=N, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                 save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3, bucket_mb=64)
diff.kl_weight = 0.1
losses = []
for i, (b, c) in enumerate(batches):
    torch.manual_seed(100 + i)
    np.random.seed(200 + i)
    loop.forward_backward(b, c)
    loop.optimize_normal()
    losses.append(float(loop.last_losses["loss"].mean()))
torch.cuda.synchronize()
f = loop.opt.flat.flat
rec = {"mode": mode, "losses": losses, "sum": float(f.double().sum()), "sumsq": float((f.double() ** 2).sum()), "absmax": float(f.abs().max()),
       "probe": f[::100003].double().cpu().tolist(), "buckets": len(loop.buckets.buckets), "collectives": loop.buckets.launched,
       "active": bool(loop.buckets.active), "backend": dist.get_backend() if dist.is_initialized() else None}
print("RCCL1 " + json.dumps(rec))
if dist.is_initialized():
    dist.destroy_process_group()
