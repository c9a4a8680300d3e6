"""Dev tool: the 15 ResBlock entry sweeps of a P64 DDIM step at batch 128 (1x1 skip conv + first GroupNorm written as f16 planes,
ops.skip_gn_fused) timed one shape at a time with HIP events, against the two roofs that bound them: HBM (read x fp32, write the
planes, write the skip) and f16x3 MFMA.  Shapes: reference unet.py:165-171 (skip_connection) for every block whose channel count
changes, unet.py:386-470 with channel_mult (1,2,3,4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops
B = 128
# (C1 = channels of h, C2 = channels of the skip-stack entry (0: encoder block), Cout, res, count per step)
SHAPES = [(128, 128, 128, 64, 2), (256, 128, 128, 64, 1),
          (128, 0, 256, 32, 1), (256, 128, 256, 32, 1), (256, 256, 256, 32, 1), (384, 256, 256, 32, 1),
          (256, 0, 384, 16, 1), (384, 256, 384, 16, 1), (384, 384, 384, 16, 1), (512, 384, 384, 16, 1),
          (384, 0, 512, 8, 1), (512, 384, 512, 8, 1), (512, 512, 512, 8, 2)]
if __name__ == "__main__":
    reps = 5
    tot = 0.0
    for (c1, c2, co, r, cnt) in SHAPES:
        dev = "cuda:0"
        a = ops.to_nhwc(torch.randn(B, c1, r, r, device=dev))
        x = ops.CatAct(a, ops.to_nhwc(torch.randn(B, c2, r, r, device=dev))) if c2 else a
        C = c1 + c2
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        w = (torch.randn(co, C, 1, 1, device=dev) / C ** .5)
        b = torch.randn(co, device=dev)
        with torch.no_grad():
            lz = ops.group_norm_lazy(x, gamma, beta, None, True, 32, 1e-5)
            assert ops.skip_gn_ok(lz, w)
            ops.skip_gn_fused(lz, w, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.skip_gn_fused(lz, w, b)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        M = B * r * r
        by = 4.0 * M * (2 * C + co)
        fl = 2.0 * M * C * co
        tot += us * cnt
        print(f"skip+GN {c1:3d}+{c2:3d}->{co:3d} @{r:2d}x{r:<2d} x{cnt}/step {us:7.1f} us  {by / us / 1e6:5.2f} TB/s of 8  {fl / us / 1e6:6.1f} TF of 833  "
              f"(hbm floor {by / 8e6:6.1f} us, mfma floor {fl / 833.3e6:6.1f} us)")
        del a, x, lz
    print(f"per DDIM step: {tot / 1e3:.2f} ms")
