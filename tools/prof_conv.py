"""Dev tool: a few launches of ONE conv3x3 shape on pre-split planes, for rocprofv3 passes (tools/prof_conv.sh).
usage: prof_conv.py Cin Cout res [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops
from causaldiffae_amd._lib import check, lib, ptr, stream
ci, co, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 128
x = ops.to_nhwc(torch.randn(B, ci, r, r, device="cuda:0"))
planes = torch.empty((2, B, r, r, ci), dtype=torch.float16, device="cuda:0")
check(lib.cdae_split_f16(ptr(x), ptr(planes[0]), ptr(planes[1]), x.numel(), stream()))
xs = ops.SplitAct(planes[0], planes[1], (B, ci, r, r))
w = (torch.randn(co, ci, 3, 3, device="cuda:0") / (9 * ci) ** .5).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(6):
        ops.conv3x3_ps(xs, w, None)
torch.cuda.synchronize()
