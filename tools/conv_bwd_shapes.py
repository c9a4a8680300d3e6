"""Dev tool: time fwd / dgrad / wgrad of the main conv3x3 shapes through the C-ABI (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd._lib import lib, ptr, stream, splitk_ws, SPLITK_BYTES, check
DEV = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def timeit(f, n=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ws = splitk_ws(DEV)
tot = [0, 0, 0]
for (ci, co, r, cnt) in [(128, 128, 64, 7), (256, 128, 64, 2), (384, 128, 64, 1), (256, 256, 32, 6), (512, 256, 32, 1), (640, 256, 32, 1), (384, 384, 16, 6),
                         (768, 384, 16, 1), (512, 512, 8, 10), (1024, 512, 8, 2)]:
    x = torch.randn(B, r, r, ci, device=DEV); w = torch.randn(co, 3, 3, ci, device=DEV) * 0.01
    y = torch.empty(B, r, r, co, device=DEV); dy = torch.randn(B, r, r, co, device=DEV)
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(co, device=DEV)
    sn, sy, sx, sc = r * r * ci, r * ci, ci, 1
    fl = 2.0 * B * r * r * co * 9 * ci
    t_f = timeit(lambda: check(lib.cdae_conv3x3_fwd(ptr(x), sn, sy, sx, sc, ptr(w), None, None, None, ptr(y), co, 0, B, r, r, ci, co, 1, 0, ptr(ws), SPLITK_BYTES, stream())))
    t_d = timeit(lambda: check(lib.cdae_conv3x3_dgrad(ptr(dy), co, ptr(w), ptr(dx), ci, B, r, r, ci, co, 1, 0, 0, ptr(ws), SPLITK_BYTES, stream())))
    t_w = timeit(lambda: check(lib.cdae_conv3x3_wgrad(ptr(x), sn, sy, sx, sc, ptr(dy), co, ptr(dw), None, B, r, r, ci, co, 1, 0, 0, ptr(ws), SPLITK_BYTES, stream())))
    print(f"Cin{ci:5d} Cout{co:4d} res{r:3d} x{cnt:2d}: fwd {t_f:7.3f} ms {fl/t_f/1e9:6.1f} TF | dgrad {t_d:7.3f} ms {fl/t_d/1e9:6.1f} TF | wgrad {t_w:7.3f} ms {fl/t_w/1e9:6.1f} TF")
    tot[0] += t_f * cnt; tot[1] += t_d * cnt; tot[2] += t_w * cnt
print("weighted totals (ms): fwd %.2f dgrad %.2f wgrad %.2f" % tuple(tot))
