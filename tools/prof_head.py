"""Dev tool: the UNet output head (GroupNorm -> SiLU -> conv3x3 to 4 channels, P64 batch 128) on the exact-fp32 kernel and on the
path it replaces (GroupNorm planes + plane GEMM), HIP-event timings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops
dev = "cuda:0"
for (N, C, Cout, S) in [(128, 128, 4, 64), (128, 128, 3, 64), (256, 128, 1, 32)]:
    x = ops.to_nhwc(torch.randn(N, C, S, S, device=dev))
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    w = (torch.randn(Cout, C, 3, 3, device=dev) / (3 * C ** .5)).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    with torch.no_grad():
        lz = ops.group_norm_lazy(x, gamma, beta, None, True, 32, 1e-5)
        from causaldiffae_amd._lib import check, lib
        def scalar():
            check(lib.cdae_tune_set(3, 0)); ops.head_conv(lz, w, b); check(lib.cdae_tune_set(3, 1))
        for name, fn in [("head, 4x4x1 MFMA", lambda: ops.head_conv(lz, w, b)), ("head, scalar form", scalar), ("planes + GEMM", lambda: ops.conv3x3_ps(lz.planes(), w, b, out_nchw=True))]:
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 5
            print(f"N={N} C={C}->{Cout} @{S}x{S}  {name:18s} {us:8.1f} us   (input read floor {4.0 * N * S * S * C / 8e6:6.1f} us at 8 TB/s)")
