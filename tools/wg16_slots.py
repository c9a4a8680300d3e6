"""Dev tool: wg16 (cdae_linear_wgrad_io on bf16 rows, incl. its finish launch) by the block target of its row split (tune key wg16_slots)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd._lib import check, lib, ptr, stream, tune_scope
from causaldiffae_amd.ops16 import _sk
dev = torch.device("cuda:0")
SHAPES = [(65536, 256, 256), (65536, 768, 256), (16384, 256, 256), (16384, 768, 256), (262144, 128, 128), (262144, 256, 128), (4096, 256, 2304), (8192, 384, 384), (32768, 256, 256)]
ws, wsb = _sk(dev)
for (M, N, K) in SHAPES:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    row = []
    for slots in (512, 384, 256, 192, 128, 64):
        with tune_scope(wg16_slots=slots, rows16_min_m=1):
            go = lambda: check(lib.cdae_linear_wgrad_io(ptr(x), K, ptr(dy), N, ptr(dw), K, ptr(db), M, N, K, 12, 0, ws, wsb, stream()))
            for _ in range(3): go()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): go()
            e1.record(); torch.cuda.synchronize()
            row.append("%d: %6.1f" % (slots, e0.elapsed_time(e1) * 50))
    print(f"rows={M:6d} N={N:4d} K={K:4d} | " + " | ".join(row), flush=True)
