import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from causaldiffae_amd import ops, _lib
DEV = "cuda:0"
def run(N, Cin, Cout, H, stride, splitk=True):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(N, Cin, H, H, generator=g) - .5
    w = (torch.rand(Cout, Cin, 3, 3, generator=g) - .5) / (9*Cin)**.5
    xd = ops.to_nhwc(x.to(DEV)).requires_grad_(True)
    wd = w.contiguous(memory_format=torch.channels_last).to(DEV).requires_grad_(True)
    if not splitk:
        old = _lib.SPLITK_BYTES; ops.SPLITK_BYTES = 0
        ops._sk = lambda dev: (None, 0)
    y = ops.conv3x3(xd, wd, None, stride=stride)
    xc = x.double().requires_grad_(True); wc = w.double().requires_grad_(True)
    yc = F.conv2d(xc, wc, None, stride=stride, padding=1)
    gy = torch.rand(*yc.shape, generator=g) - .5
    (y * gy.to(DEV)).sum().backward(); (yc * gy.double()).sum().backward()
    d = (xd.grad.cpu().double() - xc.grad).abs()
    bad = (d > 1e-4).nonzero()
    print(f"N{N} Cin{Cin} Cout{Cout} H{H} s{stride} splitk={splitk}: y err {(y.cpu().double()-yc).abs().max():.2e} dx err {d.max():.2e} nbad {bad.shape[0]} {bad[:6].tolist()}  dw err {(wd.grad.cpu().double()-wc.grad).abs().max():.2e}")
for sk in (True, False):
    run(4, 16, 32, 32, 2, sk)
    run(4, 16, 32, 48, 2, sk)
    run(4, 16, 32, 32, 1, sk)
    run(4, 32, 32, 16, 2, sk)
    run(2, 128, 128, 32, 2, sk)
    run(4, 4, 16, 64, 2, sk)
