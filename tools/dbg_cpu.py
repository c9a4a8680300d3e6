import os, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
from oracle import unet_ref as U
from oracle.closed_form import fill_state_dict, synth
cfg = U.default_cfg(image_size=64, in_channels=4, n_vars=4, rep_cond=True, causal_modeling=True)
sd = fill_state_dict(U.param_spec(cfg))
x = synth("bench.cpu.x", (4, 4, 64, 64)); z = synth("bench.cpu.z", (4, 512)); t = torch.full((4,), 500.0)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        U.unet_forward(sd, cfg, x, t, z=z)
        t0 = time.perf_counter(); U.unet_forward(sd, cfg, x, t, z=z); dt = time.perf_counter() - t0
    print(f"threads {nt}: batch 4 forward {dt:.2f}s -> {4/dt:.2f} img-steps/s", flush=True)
