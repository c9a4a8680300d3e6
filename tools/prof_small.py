"""Dev tool: HIP-event timings of small element-wise kernels of the training step at their model shapes (C64, batch 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd._lib import check, lib, ptr, stream
dev = "cuda:0"
for (N, H, W, C) in [(32, 32, 32, 256), (32, 16, 16, 384), (32, 8, 8, 512)]:
    src = torch.randn(N, 2 * H, 2 * W, C, device=dev)
    dst = torch.empty(N, H, W, C, device=dev)
    f = lambda: check(lib.cdae_sumpool2(ptr(src), ptr(dst), N, H, W, C, stream()))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"sumpool2 [{N}, {2*H}, {2*W}, {C}] -> [{N}, {H}, {W}, {C}]: {us:.1f} us = {(src.numel() + dst.numel()) * 4 / us / 1e6:.2f} TB/s")
from causaldiffae_amd import ops
for (N, Cin, Cout, S) in [(128, 4, 128, 64), (32, 3, 128, 64), (256, 1, 128, 32)]:
    x = torch.randn(N, Cin, S, S, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / 6
    b = torch.randn(Cout, device=dev)
    with torch.no_grad():
        f = lambda: ops.conv3x3(x, w, b)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"stem conv N={N} {Cin}->{Cout} @{S}x{S}: {us:.1f} us (write floor {4.0 * N * S * S * Cout / 8e6:.1f} us at 8 TB/s)")
