"""Dev tool: every contraction launch of one C64 batch-32 training step with its shape, tile and split (the library's per-launch
profiling tags, CDAE_PROF_DUMP), folded by label: launches per step, average us, TFLOP/s.  Launches are timed with HIP events on the
launch stream, back to back as in the real step."""
import os, sys, collections, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = os.path.join(tempfile.gettempdir(), "cdae_prof_dump.tsv")
if os.path.exists(dump):
    os.remove(dump)
os.environ["CDAE_PROF_DUMP"] = dump
import torch
import bench
from causaldiffae_amd import _lib
from improved_diffusion import script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop
dev = torch.device("cuda:0")
M32 = os.environ.get("MODEL", "c64") == "m32"          # MODEL=m32 BATCH=256 FP16=1: BASELINE config [1] on the reduced-precision torso
B = int(os.environ.get("BATCH", "256" if M32 else "32"))
SZ, CH, NV = (32, 1, 2) if M32 else (64, 3, 4)
cfg = {**su.model_and_diffusion_defaults(), "image_size": SZ, "in_channels": CH, "n_vars": NV, "rep_cond": True, "causal_modeling": True, "class_cond": M32}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 4321)
model.to(dev).train()
data = load_data(data_dir="synthetic", batch_size=B, image_size=SZ, in_channels=CH, n_vars=NV, seed=0, device=dev, class_cond=M32)
loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=B, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                 save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=NV, causal_modeling=True, in_channels=CH,
                 use_fp16=os.environ.get("FP16") == "1")
diff.kl_weight = 0.1
for _ in range(3):
    b, c = next(data); loop.forward_backward(b, c); loop.optimize_normal()
torch.cuda.synchronize()
STEPS = 3
_lib.prof_enable(True)
for _ in range(STEPS):
    b, c = next(data); loop.forward_backward(b, c); loop.optimize_normal()
_lib.prof_read()
_lib.prof_enable(False)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for line in open(dump):
    fam, us, work, tag = line.rstrip("\n").split("\t")
    a = agg[(fam, tag)]
    a[0] += 1; a[1] += float(us); a[2] += float(work)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows) / STEPS
print(f"total {tot / 1e3:.2f} ms per step in tagged families")
for (fam, tag), (n, us, work) in rows[:int(os.environ.get("TOP", "70"))]:
    print(f"{us / STEPS / 1e3:7.3f} ms/step {n / STEPS:5.1f}x {us / n:8.1f} us {work / us / 1e6 if us else 0:7.1f} TF  fam{fam} {tag}")
