"""Dev tool: which Python call sites issue ATen kernels / hipMemcpy calls inside ONE eager P64 DDIM step (batch 128) — the step the public
loop captures into its graph.  torch.profiler CPU op tree: every outermost aten:: op that launches a kernel or a memcpy, folded by the
innermost stack frame inside this repo."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from improved_diffusion import script_util as su

dev = torch.device("cuda:0")
N = int(os.environ.get("N", "128"))
cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 4, "n_vars": 4, "rep_cond": True, "causal_modeling": True, "timestep_respacing": "ddim100"}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 1234)
model.to(dev).eval()
x = torch.randn(N, 4, 64, 64, device=dev)
z = torch.randn(N, 512, device=dev)
tab = diff._step_table(dev, N)
with torch.no_grad():
    for k in range(3):
        x = diff.ddim_sample(model, x, tab[k], model_kwargs=dict(z=z))["sample"]
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        x = diff.ddim_sample(model, x, tab[3], model_kwargs=dict(z=z))["sample"]
        torch.cuda.synchronize()
VIEWS = {"aten::view", "aten::reshape", "aten::permute", "aten::slice", "aten::select", "aten::as_strided", "aten::expand", "aten::empty", "aten::empty_like",
         "aten::empty_strided", "aten::t", "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::_unsafe_view",
         "aten::result_type", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::detach_", "aten::resize_",
         "aten::unflatten", "aten::flatten", "aten::chunk", "aten::split", "aten::narrow", "aten::unbind", "aten::stride", "aten::size", "aten::numel"}
cnt = collections.Counter()
for e in prof.events():
    if not (e.name.startswith("aten::") or "emcpy" in e.name or "emset" in e.name):
        continue
    if e.name in VIEWS:
        continue
    p, chain = e.cpu_parent, []
    while p is not None:
        chain.append(p.name)
        p = p.cpu_parent
    if any(n.startswith("aten::") and n not in VIEWS for n in chain):
        continue                                        # outermost computing aten op only
    st = [s_ for s_ in (e.stack or []) if "causaldiffae_amd" in s_ or "bench" in s_]
    cnt[(e.name, st[0][-80:] if st else "", str(e.input_shapes)[:60])] += 1
for k, c in cnt.most_common(80):
    print(c, *k, sep=" | ")
print("---- device kernels that are not libcdae's")
kc = collections.Counter()
for e in prof.events():
    if e.device_type is not None and str(e.device_type).endswith("CUDA") and ("at::native" in e.name or "rocclr" in e.name or "emcpy" in e.name.lower()):
        kc[e.name[:110]] += 1
for k, c in kc.most_common(40):
    print(c, k, sep=" | ")
