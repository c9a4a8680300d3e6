"""Dev tool: every tagged launch of one eager P64 batch-128 DDIM step with its shape, tile and split (CDAE_PROF_DUMP), folded by label."""
import os, sys, collections, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = os.path.join(tempfile.gettempdir(), "cdae_prof_dump_ddim.tsv")
if os.path.exists(dump):
    os.remove(dump)
os.environ["CDAE_PROF_DUMP"] = dump
import torch
import bench
from causaldiffae_amd import _lib
from improved_diffusion import script_util as su
dev = torch.device("cuda:0")
cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 4, "n_vars": 4, "rep_cond": True, "causal_modeling": True,
       "timestep_respacing": "ddim100"}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 1234)
model.to(dev).eval()
N = int(os.environ.get("BATCH", "128"))
x = torch.randn(N, 4, 64, 64, device=dev)
kw = dict(z=torch.randn(N, 512, device=dev))
tab = diff._step_table(dev, N)
STEPS = 3
with torch.no_grad():
    for k in range(2):
        x = diff.ddim_sample(model, x, tab[k], model_kwargs=kw)["sample"]
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    for k in range(STEPS):
        x = diff.ddim_sample(model, x, tab[2 + k], model_kwargs=kw)["sample"]
    _lib.prof_read()
    _lib.prof_enable(False)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for line in open(dump):
    fam, us, work, tag = line.rstrip("\n").split("\t")
    a = agg[(fam, tag)]
    a[0] += 1; a[1] += float(us); a[2] += float(work)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f"total {sum(v[1] for _, v in rows) / STEPS / 1e3:.2f} ms per step in tagged families")
for (fam, tag), (n, us, work) in rows[:int(os.environ.get("TOP", "60"))]:
    print(f"{us / STEPS / 1e3:7.3f} ms/step {n / STEPS:5.1f}x {us / n:8.1f} us {work / us / 1e6 if us else 0:7.1f} T  fam{fam} {tag}")
