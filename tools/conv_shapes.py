"""Dev tool: time every distinct conv3x3 / 1x1 shape of the P64 UNet forward at a given batch (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import Counter
from causaldiffae_amd import ops
from oracle import unet_ref as U
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = U.default_cfg(image_size=64, in_channels=4, n_vars=4, rep_cond=True, causal_modeling=True)
a = U.arch(cfg)
shapes = Counter()
res = 64
def visit(layers, res):
    for l in layers:
        if l[0] == "res":
            shapes[("c3", l[1], l[2], res, 1, 0)] += 1
            shapes[("c3", l[2], l[2], res, 1, 0)] += 1
            if l[1] != l[2]: shapes[("c1", l[1], l[2], res, 1, 0)] += 1
        elif l[0] == "attn":
            shapes[("c1", l[1], 3 * l[1], res, 1, 0)] += 1
            shapes[("c1", l[1], l[1], res, 1, 0)] += 1
        elif l[0] == "down":
            shapes[("c3", l[1], l[1], res, 2, 0)] += 1; res //= 2
        elif l[0] == "up":
            shapes[("c3", l[1], l[1], res, 1, 1)] += 1; res *= 2
    return res
for ls in a["input"][1:]: res = visit(ls, res)
res = visit(a["middle"], res)
for ls in a["output"]: res = visit(ls, res)
tot_t = tot_f = 0
rows = []
for (kind, ci, co, r, s, up), cnt in sorted(shapes.items()):
    x = ops.to_nhwc(torch.randn(B, ci, r, r, device=DEV))
    if kind == "c3":
        w = (torch.randn(co, ci, 3, 3, device=DEV) / (9 * ci) ** .5).contiguous(memory_format=torch.channels_last)
        f = lambda: ops.conv3x3(x, w, None, stride=s, up=bool(up))
        ro = r * 2 if up else (r - 1) // s + 1
        fl = 2.0 * B * ro * ro * co * 9 * ci
    else:
        w = torch.randn(co, ci, 1, 1, device=DEV) / ci ** .5
        f = lambda: ops.conv1x1(x, w, None)
        fl = 2.0 * B * r * r * co * ci
    with torch.no_grad():
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): f()
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    rows.append((kind, ci, co, r, s, up, cnt, ms, fl / ms / 1e9))
    tot_t += ms * cnt; tot_f += fl * cnt
for r_ in rows:
    print("%s Cin%4d Cout%4d res%3d s%d up%d x%d  %8.3f ms  %6.1f TF/s  share %.1f%%" % (*r_, 100 * r_[7] * r_[6] / tot_t))
print(f"batch {B}: total {tot_t:.2f} ms/step in convs, {tot_f/tot_t/1e9:.1f} TF/s aggregate")
