"""Dev tool (a -DCW_DEV=1 library as causaldiffae_amd/libcdae.so, CDAE_PS_DBG=32): convwin_kernel's cycle stamps — per wave, summed over the block's tiles:
top-of-tile wait (operands of the tile landed), mid-step vmcnt waits, barriers, epilogue, total — for a DDIM f16x3 shape and a bf16-row torso shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops, ops16
from causaldiffae_amd._lib import check, lib, ptr, stream, splitk_ws, precision_scope
dev = torch.device("cuda:0")
ws = splitk_ws(dev)

def show(label):
    torch.cuda.synchronize()
    t = ws.view(torch.int64)[: 64 * 4 * 8].reshape(64, 4, 8).cpu().double()
    m = t.mean(dim=(0, 1))
    print("%s | per wave, cycles: top-of-tile wait %.0f | vmcnt waits %.0f | barriers %.0f | epilogue %.0f | total %.0f  (x %.0f ns)" % (label, m[0], m[1], m[2], m[3], m[4], 10.0), flush=True)

with torch.no_grad():
    for (B, ci, co, r) in ((128, 128, 128, 64), (128, 256, 256, 32), (128, 512, 512, 8)):
        x = ops.to_nhwc(torch.randn(B, ci, r, r, device=dev))
        xs = ops.group_norm_lazy(x, torch.ones(ci, device=dev), torch.zeros(ci, device=dev), None, True, 32, 1e-5).planes(gm=True)
        w = (torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** .5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device=dev)
        for _ in range(2): ops.conv3x3_ps(xs, w, b, res=None, gn_stats=True)
        ws.zero_(); ops.conv3x3_ps(xs, w, b, res=None, gn_stats=True)
        show("f16x3 %d->%d @%dx%d N=%d" % (ci, co, r, r, B))
    with precision_scope("mixed16"):
        for (N, ci, co, S) in ((256, 128, 128, 32), (256, 256, 256, 16)):
            x = torch.randn(N, S, S, ci, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
            w = (torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** .5).contiguous(memory_format=torch.channels_last)
            b = torch.randn(co, device=dev)
            f = lambda: ops16._conv_fwd(x, w, b, None, N, S, S, ci, co, stream(), want_parts=True)
            f(); f(); ws.zero_(); f()
            show("bf16 rows %d->%d @%dx%d N=%d" % (ci, co, S, S, N))
