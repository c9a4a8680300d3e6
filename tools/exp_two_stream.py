"""Dev experiment: one P64 DDIM step at batch 128 as ONE chain of launches vs as TWO half-batch chains on two streams (captured into one
HIP graph each way): do the latency-bound low-resolution kernels of one half fill the slots the other half leaves empty?"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
import bench
from causaldiffae_amd._lib import ws_lane
from improved_diffusion import script_util as su

dev = th.device("cuda:0")
N = int(os.environ.get("N", "128"))
PARTS = int(os.environ.get("PARTS", "2"))
cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 4, "n_vars": 4, "rep_cond": True, "causal_modeling": True, "timestep_respacing": "ddim100"}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 1234)
model.to(dev).eval()
x = th.randn(N, 4, 64, 64, device=dev)
z = th.randn(N, 512, device=dev)
steps = diff._step_table(dev, N)
img, out, pred, t = x.clone(), th.empty_like(x), th.empty_like(x), th.empty(N, dtype=th.int64, device=dev)
side = [th.cuda.Stream() for _ in range(PARTS)]
bounds = [(i * N // PARTS, (i + 1) * N // PARTS) for i in range(PARTS)]


def body(parts):
    if parts == 1:
        eps = diff._model_eps(model, img, t, dict(z=z))
        diff._fused_update(True, img, eps, t, True, 0.0, None, sample_out=out, pred_out=pred)
    else:
        cur = th.cuda.current_stream()
        for i, (lo, hi) in enumerate(bounds):
            s = side[i]
            s.wait_stream(cur)
            with th.cuda.stream(s), ws_lane(1 + i):
                eps = diff._model_eps(model, img[lo:hi], t[lo:hi], dict(z=z[lo:hi]))
                diff._fused_update(True, img[lo:hi], eps, t[lo:hi], True, 0.0, None, sample_out=out[lo:hi], pred_out=pred[lo:hi])
        for s in side:
            cur.wait_stream(s)
    img.copy_(out)


def capture(parts):
    s = th.cuda.Stream()
    s.wait_stream(th.cuda.current_stream())
    with th.cuda.stream(s):
        body(parts); body(parts)
    th.cuda.current_stream().wait_stream(s)
    g = th.cuda.CUDAGraph()
    with th.cuda.graph(g):
        body(parts)
    return g


with th.no_grad():
    res = {}
    for parts in (1, PARTS, 1, PARTS):
        img.copy_(x); t.copy_(steps[0])
        g = capture(parts)
        for k in range(5):
            t.copy_(steps[k]); g.replay()
        ts = []
        for r in range(3):
            th.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(30):
                t.copy_(steps[(5 + k) % 100]); g.replay()
            th.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 30 * 1e3)
        print(f"parts={parts}: {statistics.median(ts):.3f} ms/step  {[round(v, 3) for v in ts]}", flush=True)
    # same result?  (different tile / split choices per half: not bit-identical, but within the parity bar)
    img.copy_(x); t.copy_(steps[0]); body(1); a = out.clone()
    img.copy_(x); t.copy_(steps[0]); body(PARTS); th.cuda.synchronize(); b = out.clone()
    print("max |one chain - split chains| =", (a - b).abs().max().item())
