"""Dev tool: the 1x1-conv / linear GEMMs of the C64 training step (fp32 operands, in-kernel split) timed for forced tile sizes and
K splits (CDAE_GEMM_DEV=1 lets the dispatcher take CDAE_TILE_FORCE / CDAE_KS_FORCE per call) against the dispatcher's own choice."""
import os, sys
os.environ["CDAE_GEMM_DEV"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops
from causaldiffae_amd._lib import check, lib, ptr, stream
dev = torch.device("cuda:0")
# (kind, rows, out features N, in features K, launches per step)   y[rows][N] = x[rows][K] w[N][K]^T
SHAPES = [("wgrad", 131072, 128, 128, 5), ("dgrad", 131072, 128, 128, 5), ("wgrad", 32768, 256, 256, 4), ("dgrad", 32768, 256, 256, 4),
          ("fwd", 2048, 512, 512, 6), ("dgrad", 2048, 512, 512, 11), ("wgrad", 2048, 512, 512, 11),
          ("fwd", 2048, 1536, 512, 6), ("dgrad", 2048, 1536, 512, 6), ("wgrad", 2048, 1536, 512, 6),
          ("fwd", 8192, 384, 384, 5), ("dgrad", 8192, 384, 384, 9), ("wgrad", 8192, 384, 384, 9),
          ("fwd", 8192, 1152, 384, 5), ("dgrad", 8192, 1152, 384, 5), ("wgrad", 8192, 1152, 384, 5)]
KS = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256]


def timed(f, n=8):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for kind, M, N, K, cnt in SHAPES:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** .5; dy = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    ws, wsb = ops._sk(dev)
    st = stream()
    if kind == "fwd":
        f = lambda: check(lib.cdae_linear_fwd(ptr(x), K, ptr(w), K, None, None, None, ptr(y), N, None, None, M, N, K, 1.0, 0, ws, wsb, st))
    elif kind == "dgrad":
        f = lambda: check(lib.cdae_linear_dgrad(ptr(dy), N, ptr(w), K, ptr(dx), K, M, N, K, 0, ws, wsb, st))
    else:
        f = lambda: check(lib.cdae_linear_wgrad(ptr(x), K, ptr(dy), N, ptr(dw), K, ptr(db), M, N, K, 0, ws, wsb, st))
    os.environ.pop("CDAE_TILE_FORCE", None); os.environ.pop("CDAE_KS_FORCE", None)
    base = timed(f)
    if os.environ.get("ONLY_DEFAULT"):
        print(f"{kind:5s} rows={M:6d} N={N:4d} K={K:4d} x{cnt:2d}: default {base:6.1f} us")
        continue
    res = []
    for tile in (64, 128):
        for ks in KS:
            os.environ["CDAE_TILE_FORCE"] = str(tile); os.environ["CDAE_KS_FORCE"] = str(ks)
            try:
                res.append((timed(f), tile, ks))
            except Exception as e:
                pass
    res.sort()
    print(f"{kind:5s} rows={M:6d} N={N:4d} K={K:4d} x{cnt:2d}: default {base:6.1f} us | best " + "  ".join(f"{t:5.1f}us t{tile} ks{ks}" for t, tile, ks in res[:4]))
