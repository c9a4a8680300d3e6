"""Dev tool: the window wgrad kernel (+ its split-K finish) on every stride-1 conv3x3 shape of the C64 training step at batch 32,
HIP-event timings and the fraction of the bf16x3 roof (833 TFLOP/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops
from causaldiffae_amd._lib import check, lib, ptr, stream
from prof_shapes import SHAPES
B = int(os.environ.get("B", "32"))
dev = torch.device("cuda:0")
tot = 0.0
for (ci, co, r, cnt) in SHAPES:
    a = torch.randn(2, B, r, r, ci, device=dev).to(torch.bfloat16)
    d = torch.randn(2, B, r, r, co, device=dev).to(torch.bfloat16)
    dw = torch.zeros(co, 3, 3, ci, device=dev)
    db = torch.zeros(co, device=dev)
    ws, wsb = ops._sk(dev)
    st = stream()
    def run():
        check(lib.cdae_conv3x3_wgrad_win(ptr(a[0]), ptr(a[1]), ptr(d[0]), ptr(d[1]), ptr(dw), ptr(db), B, r, r, ci, co, 1, ws, wsb, st))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    fl = 2.0 * B * r * r * co * ci * 9
    tot += us * cnt
    print(f"wgrad {ci:4d}->{co:3d} @{r:2d}x{r:<2d} x{cnt:2d}/step  {us:8.1f} us  {fl / us / 1e6:6.1f} TF  frac {fl / us / 1e6 / 833.3:.3f}")
print(f"sum over the step's launches: {tot / 1e3:.2f} ms")
