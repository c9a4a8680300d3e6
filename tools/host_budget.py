"""Dev: where the host CPU of a training step goes.  Per-thread CPU time (/proc/self/task), wall and CPU per step at the benchmark batch
(GPU-bound) and at batch 2 (host-bound: the wall time there is the host's critical path), then a cProfile of the host-bound step."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from improved_diffusion import script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop
import bench

dev = torch.device("cuda:0")
if os.environ.get("SETDEV") == "1":
    torch.cuda.set_device(0)
if os.environ.get("AVAIL") == "1":
    assert torch.cuda.is_available() and torch.cuda.device_count() >= 1
TICK = os.sysconf("SC_CLK_TCK")


def threads():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / TICK)
        except OSError:
            pass
    return out


def make(batch):
    cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True}
    model, diff = su.create_model_and_diffusion(**cfg)
    bench.randomize(model, 4321)
    model.to(dev).train()
    data = load_data(data_dir="synthetic", batch_size=batch, image_size=64, in_channels=3, n_vars=4, seed=0, device=dev)
    loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=batch, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                     save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3)
    diff.kl_weight = 0.1

    def steps(n):
        for _ in range(n):
            b, c = next(data)
            loop.forward_backward(b, c)
            loop.optimize_normal()
        torch.cuda.synchronize()
    return steps


for batch in (32, 2) if os.environ.get("ONLY32") != "1" else (32,):
    steps = make(batch)
    steps(3)
    t0, c0, th0 = time.perf_counter(), time.process_time(), threads()
    n = 20
    steps(n)
    t1, c1, th1 = time.perf_counter(), time.process_time(), threads()
    print(f"batch {batch}: wall {1e3 * (t1 - t0) / n:.2f} ms/step, process cpu {1e3 * (c1 - c0) / n:.2f} ms/step")
    for tid, (name, cpu) in sorted(th1.items(), key=lambda kv: -(kv[1][1] - th0.get(kv[0], ("", 0))[1])):
        d = cpu - th0.get(tid, ("", 0))[1]
        if d > 0.005:
            print(f"    thread {tid} {name}: {1e3 * d / n:.2f} ms/step")

if os.environ.get("ONLY32") == "1":
    sys.exit(0)
torch.autograd.set_multithreading_enabled(False)
steps(2)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); steps(10); pr.disable()
print("batch 2, autograd on the main thread, under cProfile: wall ms/step", 100 * (time.perf_counter() - t0))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
