"""Dev tool: cdae_gemm16_ps on the torso's 1 x 1 conv / linear shapes — the streaming kernel (rows16.hip) beside the
plane GEMM, HIP-event timing.   python3 tools/rows16_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd._lib import check, lib, ptr, stream, tune_scope
from causaldiffae_amd.ops16 import _sk

dev = torch.device("cuda:0")
SHAPES = [(65536, 256, 256, 0), (65536, 256, 256, 1), (65536, 768, 256, 0), (65536, 256, 768, 0), (262144, 128, 128, 0), (262144, 128, 128, 1),
          (16384, 256, 256, 0), (16384, 256, 256, 1), (16384, 768, 256, 0), (16384, 256, 768, 0), (4096, 256, 256, 0), (8192, 384, 384, 0), (8192, 384, 384, 1), (8192, 1152, 384, 0), (2048, 512, 512, 0), (2048, 512, 512, 1), (2048, 1536, 512, 0)]
ws, wsb = _sk(dev)


def run(M, N, K, res, reps=30):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def go():
        check(lib.cdae_gemm16_ps(ptr(a), K, ptr(w), K, ptr(b), ptr(r), ptr(c), N, None, M, N, K, 3 if res else 1, 0, ws, wsb, stream()))
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (M, N, K, res) in SHAPES:
    byts = 2.0 * M * K + 2.0 * M * N * (2 if res else 1) + 2.0 * N * K
    row = []
    for cfg in (dict(rows16_min_m=1 << 30), dict(rows16_min_m=1, rows16_ring=0), dict(rows16_min_m=1)):
        with tune_scope(**cfg):
            row.append(run(M, N, K, res))
    print(f"M={M:6d} N={N:4d} K={K:4d} res={res}  plane {row[0]:7.1f} us   rows16 regs {row[1]:7.1f} us ({byts / row[1] / 1e6:5.2f} TB/s)  as dispatched {row[2]:7.1f} us ({byts / row[2] / 1e6:5.2f} TB/s)   hbm floor @4.5 {byts / 4.5e6:6.1f} us", flush=True)
