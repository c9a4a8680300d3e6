"""Worker of tests/test_gpu_kernels.py::test_window_conv_odd_shapes: the second-generation window kernel forced onto shapes whose
last tile is partial in M and / or N (env CDAE_CONVWIN_MINTILES=1 is read once per process, hence the subprocess)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from causaldiffae_amd import ops
from causaldiffae_amd._lib import check, lib, ptr, stream, range_check

def split(x):
    N, C, H, W = x.shape
    planes = torch.empty((2, N, H, W, C), dtype=torch.float16, device=x.device)
    check(lib.cdae_split_f16(ptr(x), ptr(planes[0]), ptr(planes[1]), x.numel(), stream()))
    return ops.SplitAct(planes[0], planes[1], (N, C, H, W))

g = torch.Generator(device="cuda:0").manual_seed(41)
worst = 0.0
for (N, ci, co, S, res) in [(5, 64, 96, 8, False),        # M = 320: second tile has 64 rows; N = 96 < 128 columns
                            (129, 512, 512, 8, True),     # odd batch at the 8 x 8 level: M % 256 = 64, four n-tiles, residual
                            (3, 128, 160, 16, False),     # M = 768 = 3 tiles of one image each; second n-tile 32 columns wide
                            (7, 96, 128, 32, True),       # Cin = 96: three 32-channel chunks
                            (1, 32, 32, 64, False),       # one image, one chunk (18 K-steps), 16 tiles
                            (2, 256, 384, 32, False)]:    # three n-tiles
    x = ops.to_nhwc(torch.randn(N, ci, S, S, device="cuda:0", generator=g))
    w = (torch.randn(co, ci, 3, 3, device="cuda:0", generator=g) / (9 * ci) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device="cuda:0", generator=g)
    r = ops.to_nhwc(torch.randn(N, co, S, S, device="cuda:0", generator=g)) if res else None
    with torch.no_grad():
        y = ops.conv3x3_ps(split(x), w, b, res=r)
    exact = F.conv2d(x.double().contiguous(), w.double(), b.double(), padding=1)
    if res:
        exact = exact + r.double()
    e = (y.double() - exact).abs().max().item() / max(1.0, exact.abs().max().item())
    print(f"N={N} {ci}->{co} @{S}: rel err {e:.2e}")
    worst = max(worst, e)
    assert torch.isfinite(y).all()
range_check("odd shapes")
assert worst < 2e-5, worst
print("ok")
