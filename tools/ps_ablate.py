"""Dev tool: time the pre-split conv kernel on representative layer shapes (env CDAE_PS_DBG selects an ablation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops
from causaldiffae_amd._lib import check, lib, ptr, stream
DEV = "cuda:0"
B = 128


def split(x):
    x = ops.to_nhwc(x)
    N, C, H, W = x.shape
    planes = torch.empty((2, N, H, W, C), dtype=torch.float16, device=x.device)
    check(lib.cdae_split_f16(ptr(x), ptr(planes[0]), ptr(planes[1]), x.numel(), stream()))
    return ops.SplitAct(planes[0], planes[1], (N, C, H, W))


STAMPS = (int(os.environ.get("CDAE_PS_DBG", "0")) & 32) != 0
for (ci, co, r) in [(128, 128, 64), (256, 256, 32), (384, 384, 16), (512, 512, 8), (1024, 512, 8)]:
    x = torch.randn(B, ci, r, r, device=DEV)
    xs = split(x)
    w = (torch.randn(co, ci, 3, 3, device=DEV) / (9 * ci) ** .5).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            ops.conv3x3_ps(xs, w, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv3x3_ps(xs, w, None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        fl = 2.0 * B * r * r * co * ci * 9
        y = ops.conv3x3_ps(xs, w, None)
        exact = torch.nn.functional.conv2d(x[:4].double(), w.double(), padding=1)
        err = (y[:4].double() - exact).abs().max().item()
        err2 = (y[-2:].double() - torch.nn.functional.conv2d(x[-2:].double(), w.double(), padding=1)).abs().max().item()
        print(f"dbg={os.environ.get('CDAE_PS_DBG','0')} cw={os.environ.get('CDAE_CONVWIN','1')} conv {ci}->{co} @{r}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF  err {max(err, err2):.2e}")
        if STAMPS and os.environ.get("CDAE_CONVWIN", "1") != "0":
            from causaldiffae_amd._lib import splitk_ws
            w64 = splitk_ws(torch.device(DEV)).view(torch.int64)[:64 * 4 * 8].reshape(64 * 4, 8).double().cpu()
            tot = w64[:, 4].mean()
            print(f"    convwin stamps (cycles per wave, mean over 256 waves): total {tot:.0f}  tile-top wait {w64[:,0].mean():.0f} ({w64[:,0].mean()/tot:.3f})  "
                  f"mid vmcnt wait {w64[:,1].mean():.0f} ({w64[:,1].mean()/tot:.3f})  mid barrier {w64[:,2].mean():.0f} ({w64[:,2].mean()/tot:.3f})  "
                  f"epilogue issue {w64[:,3].mean():.0f} ({w64[:,3].mean()/tot:.3f})")
        elif STAMPS:
            from causaldiffae_amd._lib import splitk_ws
            w64 = splitk_ws(torch.device(DEV)).view(torch.int64)[:64 * 8 * 4].reshape(64 * 8, 4).double().cpu()
            tot = w64.sum(1, keepdim=True)
            sh = (w64 / tot).mean(0)
            steps = 9 * ci // 32
            print(f"    per step (cycles @100MHz-ish memtime units): wait+barrier {w64[:,0].mean()/steps:8.1f}  issue {w64[:,1].mean()/steps:8.1f}  "
                  f"compute {w64[:,2].mean()/steps:8.1f}   shares wait {sh[0]:.2f} issue {sh[1]:.2f} compute {sh[2]:.2f} tail {sh[3]:.2f}")
