"""Dev tool: GroupNorm -> qkv (ops.linear_gn) at the two attention levels of the P64 UNet, batch 128 / 32, HIP-event timings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops
dev = "cuda:0"
for (N, C, H) in [(128, 384, 16), (128, 512, 8), (32, 384, 16), (32, 512, 8)]:
    x = torch.randn(N, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w = torch.randn(3 * C, C, device=dev) / C ** .5
    b = torch.zeros(3 * C, device=dev)
    with torch.no_grad():
        lz = ops.group_norm_lazy(x, gamma, beta, None, False, 32, 1e-5)
        def one(): return ops.linear_gn(lz, w, b)
        def two(): return ops.linear_ps(lz.planes(), w, b)
        for name, f in (("one pass", one), ("planes + plane GEMM", two)):
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            fl = 2.0 * N * H * H * C * 3 * C
            print(f"value norm->qkv N={N} C={C} {H}x{H} {name:20s} {us:7.1f} us {fl / us / 1e6:6.1f} TF")
