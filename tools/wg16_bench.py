"""Dev tool: cdae_linear_wgrad_io on bf16 rows — wg16.hip beside the general GEMM path, HIP-event timing.   python3 tools/wg16_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd._lib import check, lib, ptr, stream, tune_scope
from causaldiffae_amd.ops16 import _sk

dev = torch.device("cuda:0")
SHAPES = [(65536, 256, 256), (262144, 128, 128), (65536, 768, 256), (16384, 256, 256), (16384, 768, 256), (262144, 256, 128), (65536, 128, 256), (4096, 256, 256)]
ws, wsb = _sk(dev)


def run(M, N, K, reps=20):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    dw = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    def go():
        check(lib.cdae_linear_wgrad_io(ptr(x), K, ptr(dy), N, ptr(dw), K, ptr(db), M, N, K, 12, 0, ws, wsb, stream()))
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (M, N, K) in SHAPES:
    byts = 2.0 * M * (N + K)
    row = []
    for cfg in (dict(rows16_min_m=1 << 30), dict(rows16_min_m=1)):
        with tune_scope(**cfg):
            row.append(run(M, N, K))
    print(f"rows={M:6d} N={N:4d} K={K:4d}  general {row[0]:7.1f} us   wg16 {row[1]:7.1f} us ({byts / row[1] / 1e6:5.2f} TB/s of operands)   hbm floor @4.5 {byts / 4.5e6:6.1f} us", flush=True)
