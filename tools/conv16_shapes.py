"""Dev tool: HIP-event timings of the 16-bit torso's stride-1 conv3x3 forward (cdae_conv3x3_fwd16: bf16 rows in and out, the window kernel's
one-plane / channel-halves form) at BASELINE config [1]'s shapes (M32, batch 256) and the C64 batch-32 shapes, with the residual and the
GroupNorm partial sums the model asks for.  In a -DCW_DEV=1 build CDAE_PS_DBG=256 drops the epilogue, 4 the weight DMAs, 8 the barrier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import causaldiffae_amd
from causaldiffae_amd import ops, ops16
from causaldiffae_amd._lib import precision_scope, stream

SHAPES = [(256, 128, 128, 32), (256, 256, 256, 16), (256, 256, 256, 8), (32, 128, 128, 64), (32, 256, 256, 32), (32, 384, 384, 16)]
with precision_scope("mixed16"), torch.no_grad():
    for (N, ci, co, S) in SHAPES:
        for res in (False, True):
            x = torch.randn(N, S, S, ci, device="cuda:0").to(torch.bfloat16).permute(0, 3, 1, 2)
            w = (torch.randn(co, ci, 3, 3, device="cuda:0") / (9 * ci) ** .5).contiguous(memory_format=torch.channels_last)
            b = torch.randn(co, device="cuda:0")
            r = torch.randn(N, S, S, co, device="cuda:0").to(torch.bfloat16).permute(0, 3, 1, 2) if res else None
            st = stream()
            f = lambda: ops16._conv_fwd(x, w, b, r, N, S, S, ci, co, st, want_parts=os.environ.get("PARTS", "1") == "1")
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 5
            fl = 2.0 * N * S * S * co * ci * 9
            print(f"conv16 {ci:4d}->{co:3d} @{S:2d}x{S:<2d} N={N:3d} res={int(res)}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF  frac {fl / us / 1e6 / 2500:.3f} of 2500")
