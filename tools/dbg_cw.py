"""Dev tool: where does the convwin kernel differ from F.conv2d?  (error maps by image, tile, row, column)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops
from causaldiffae_amd._lib import check, lib, ptr, stream
B, ci, co, r = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 128, 128, 64)
torch.manual_seed(0)
x = torch.randn(B, ci, r, r, device="cuda:0")
xn = ops.to_nhwc(x)
planes = torch.empty((2, B, r, r, ci), dtype=torch.float16, device="cuda:0")
check(lib.cdae_split_f16(ptr(xn), ptr(planes[0]), ptr(planes[1]), xn.numel(), stream()))
xs = ops.SplitAct(planes[0], planes[1], (B, ci, r, r))
w = (torch.randn(co, ci, 3, 3, device="cuda:0") / (9 * ci) ** .5).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    y = ops.conv3x3_ps(xs, w, None)
    exact = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
e = (y.double() - exact).abs()            # [B, co, r, r]
print("max err", e.max().item(), "frac bad", (e > 1e-3).double().mean().item(), "nan", torch.isnan(y).sum().item())
print("per image  :", [f"{v:.1e}" for v in e.amax(dim=(1, 2, 3)).tolist()])
print("per y      :", [f"{v:.0e}" for v in e.amax(dim=(0, 1, 3)).tolist()])
print("per x      :", [f"{v:.0e}" for v in e.amax(dim=(0, 1, 2)).tolist()])
print("per channel (first 32):", [f"{v:.0e}" for v in e.amax(dim=(0, 2, 3)).tolist()[:32]])
flat = e.permute(0, 2, 3, 1).reshape(-1, co)        # [M, co]
rows = flat.amax(dim=1).reshape(-1, 256)            # per 256-row tile
print("per row-in-tile (16-row groups):", [f"{v:.0e}" for v in rows.amax(dim=0).reshape(16, 16).amax(dim=1).tolist()])
print("per tile (first 16):", [f"{v:.0e}" for v in rows.amax(dim=1).tolist()[:16]])
bad = (flat > 1e-3)
print("bad fraction per 16-col tile:", [f"{v:.2f}" for v in bad.double().mean(dim=0).reshape(-1, 16).mean(dim=1).tolist()])
