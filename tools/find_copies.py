"""Dev tool: where do the device-to-device copies of an eager DDIM step come from?  (torch profiler with stacks on one model forward)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import causaldiffae_amd  # noqa
from improved_diffusion import script_util as su
cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 4, "n_vars": 4, "rep_cond": True, "causal_modeling": True, "timestep_respacing": "ddim100"}
model, diff = su.create_model_and_diffusion(**cfg)
model.to("cuda:0").eval()
N = 16
x = torch.randn(N, 4, 64, 64, device="cuda:0"); z = torch.randn(N, 512, device="cuda:0")
t = torch.full((N,), 50, device="cuda:0", dtype=torch.long)
with torch.no_grad():
    for _ in range(2):
        diff.ddim_sample(model, x, t, model_kwargs={"z": z})
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        diff.ddim_sample(model, x, t, model_kwargs={"z": z})
        torch.cuda.synchronize()
import collections
cnt = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::") or "copy" in ev.name.lower() or "Memcpy" in ev.name)
for name, c in cnt.most_common(30):
    print(c, name)
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=12, max_name_column_width=40, max_src_column_width=110))
