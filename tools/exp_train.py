"""Dev: wall time of the C64 batch-32 training step (the bench's train leg alone), for A/B runs under environment switches:
    python tools/exp_train.py            ;  CDAE_TRAIN_GNPARTS=0 python tools/exp_train.py
Prints ms per step of REGIONS regions of STEPS steps each (defaults 3 x 40)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from improved_diffusion import script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop

dev = torch.device("cuda:0")
if os.environ.get("PTRACE_ANY"):          # a debugger started beside the job may attach (tools/spin_bt.sh)
    import ctypes
    ctypes.CDLL(None).prctl(0x59616D61, ctypes.c_ulong(-1), 0, 0, 0)
if os.environ.get("OFF"):               # OFF=name,name: fused paths of causaldiffae_amd.ops.PATH_TOGGLES switched off for this run (same-box A/B)
    from causaldiffae_amd import ops as _ops
    for _n in os.environ["OFF"].split(","):
        setattr(_ops, _ops.PATH_TOGGLES[_n], False)
if os.environ.get("SIDE_GROUP"):           # weight-gradient launches handed to the side stream in groups of this many (one cross-stream dependency per group)
    from causaldiffae_amd import ops as _ops2
    _ops2._SIDE_GROUP = int(os.environ["SIDE_GROUP"])
if os.environ.get("TUNE"):                  # TUNE=key=value,key=value: dispatch thresholds of the library (cdae_tune_set) for a same-box A/B
    from causaldiffae_amd import _lib as _l3
    for kv in os.environ["TUNE"].split(","):
        k, v = kv.split("=")
        assert _l3.lib.cdae_tune_set(_l3.TUNE_KEYS[k], int(v)) == 0
B = int(os.environ.get("BATCH", "32"))
if os.environ.get("NJ3"):
    from causaldiffae_amd._lib import lib as _l
    _l.cdae_tune_set(2, int(os.environ["NJ3"]))
STEPS, REGIONS = int(os.environ.get("STEPS", "40")), int(os.environ.get("REGIONS", "3"))
cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 4321)
model.to(dev).train()
data = load_data(data_dir="synthetic", batch_size=B, image_size=64, in_channels=3, n_vars=4, seed=0, device=dev)
loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=B, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                 save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3,
                 use_graph=bool(int(os.environ.get("GRAPH", "0"))), use_fp16=bool(int(os.environ.get("FP16", "0"))))
diff.kl_weight = 0.1


def step():
    b, c = next(data)
    loop.forward_backward(b, c)
    loop.optimize_normal()


for _ in range(8):
    step()
out, cpu = [], []
th0 = bench.thread_cpu()
for _ in range(REGIONS):
    torch.cuda.synchronize()
    c0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / STEPS * 1e3)
    cpu.append((time.process_time() - c0) / STEPS * 1e3)
th1 = bench.thread_cpu()
per_thread = sorted(((name, round(1e3 * (c - th0.get(tid, ("", 0.0))[1]) / (STEPS * REGIONS), 1)) for tid, (name, c) in th1.items()), key=lambda kv: -kv[1])
print("threads ms/step:", [kv for kv in per_thread if kv[1] >= 0.3])
print("train ms/step:", " ".join(f"{v:.3f}" for v in out), "| host cpu ms/step:", " ".join(f"{v:.1f}" for v in cpu),
      "| graphs:", len(loop._graphs), "failed:", loop._graph_failed, "| loss", float(loop.last_losses["loss"].mean()), "| env:", {k: v for k, v in os.environ.items() if k.startswith("CDAE_") or k == "OFF"})
