"""Dev tool: pre-split path vs in-kernel split on the C64 / P64 UNet forward (should be bit-identical)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from causaldiffae_amd import ops
import test_gpu_model as T

for tag in ("C64", "P64"):
    model, diff, cfg = T.make(tag)
    model.eval()
    x, x0, c, z, y = T.model_inputs(tag, cfg, 2)
    t = torch.tensor([37.0, 990.0], device="cuda:0")
    with torch.no_grad():
        e1 = model(x.cuda(), t, z=z.cuda())[0]
        orig = ops.presplit_ok
        ops.presplit_ok = lambda: False
        e0 = model(x.cuda(), t, z=z.cuda())[0]
        ops.presplit_ok = orig
    print(tag, "max|ps - ref| =", (e1 - e0).abs().max().item(), "per-channel", (e1 - e0).abs().amax(dim=(0, 2, 3)).tolist())
