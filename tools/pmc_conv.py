"""Dev tool: a few representative conv launches for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops
DEV = "cuda:0"
B = 128
for (ci, co, r, s, up) in [(128, 128, 64, 1, 0), (256, 256, 32, 1, 1), (384, 384, 16, 1, 0), (512, 512, 8, 1, 0)]:
    x = ops.to_nhwc(torch.randn(B, ci, r, r, device=DEV))
    w = (torch.randn(co, ci, 3, 3, device=DEV) / (9 * ci) ** .5).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            ops.conv3x3(x, w, None, stride=s, up=bool(up))
    torch.cuda.synchronize()
