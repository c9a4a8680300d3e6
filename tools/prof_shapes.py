"""Dev tool: every distinct stride-1 conv3x3 shape of the P64 UNet forward at batch 128 on pre-split planes, 1 warm-up + 3 launches
each, in a fixed order — the launch list behind profiles/r02_conv_shapes.md (tools/prof_shapes.sh runs it under rocprofv3 and
tools/prof_shapes_fold.py maps dispatches back to shapes).  With --time it prints HIP-event timings instead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops
from causaldiffae_amd._lib import check, lib, ptr, stream
B = int(os.environ.get("BATCH", "128"))          # BATCH=16: the reference evaluation script's batch
# (Cin, Cout, res, count per DDIM step) — reference unet.py:386-470 with channel_mult (1,2,3,4), 2 res blocks, P64
SHAPES = [(128, 128, 64, 7), (256, 128, 64, 2), (384, 128, 64, 1),
          (128, 256, 32, 1), (256, 256, 32, 6), (384, 256, 32, 1), (512, 256, 32, 1), (640, 256, 32, 1),
          (256, 384, 16, 1), (384, 384, 16, 6), (640, 384, 16, 1), (768, 384, 16, 1), (896, 384, 16, 1),
          (384, 512, 8, 1), (512, 512, 8, 10), (896, 512, 8, 1), (1024, 512, 8, 2)]
if __name__ == "__main__":
    timing = "--time" in sys.argv
    with_res = "--res" in sys.argv          # convs that carry the block's residual (conv2 of every ResBlock: half of the launches)
    gn = "--gn" in sys.argv                 # the epilogue also leaves the next GroupNorm's partial sums, as in the model
    if os.environ.get("NJ3"):               # dev: force / forbid the 96-column tiles (cdae_tune_set CONVWIN_NJ3)
        check(lib.cdae_tune_set(2, int(os.environ["NJ3"])))
    for (ci, co, r, cnt) in SHAPES:
        x = ops.to_nhwc(torch.randn(B, ci, r, r, device="cuda:0"))
        planes = torch.empty((2, B, r, r, ci), dtype=torch.float16, device="cuda:0")
        check(lib.cdae_split_f16(ptr(x), ptr(planes[0]), ptr(planes[1]), x.numel(), stream()))
        xs = ops.SplitAct(planes[0], planes[1], (B, ci, r, r))
        if "--gm" in sys.argv:           # group-major planes, as the model's GroupNorm apply / entry sweep write them for these convs
            with torch.no_grad():
                xs = ops.group_norm_lazy(x, torch.ones(ci, device="cuda:0"), torch.zeros(ci, device="cuda:0"), None, True, 32, 1e-5).planes(gm=True)
        w = (torch.randn(co, ci, 3, 3, device="cuda:0") / (9 * ci) ** .5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device="cuda:0")
        res = ops.to_nhwc(torch.randn(B, co, r, r, device="cuda:0")) if with_res else None
        with torch.no_grad():
            ops.conv3x3_ps(xs, w, b, res=res, gn_stats=gn)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.conv3x3_ps(xs, w, b, res=res, gn_stats=gn)
            e1.record()
            torch.cuda.synchronize()
        if timing:
            us = e0.elapsed_time(e1) * 1e3 / 3
            fl = 2.0 * B * r * r * co * ci * 9
            print(f"conv3x3 {ci:4d}->{co:3d} @{r:2d}x{r:<2d} x{cnt:2d}/step  {us:8.1f} us  {fl / us / 1e6:6.1f} TF  frac {fl / us / 1e6 / 833.3:.3f}")
        del x, planes, xs
