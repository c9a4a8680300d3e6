import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd._lib import lib, ptr, stream, splitk_ws, SPLITK_BYTES, check
DEV = torch.device("cuda:0"); B = 32
ws = splitk_ws(DEV)
for (ci, co, r) in [(128, 128, 64), (256, 256, 32)]:
    x = torch.randn(B, r, r, ci, device=DEV); w = torch.randn(co, 3, 3, ci, device=DEV) * 0.01
    y = torch.empty(B, r, r, co, device=DEV); dy = torch.randn(B, r, r, co, device=DEV)
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    sn, sy, sx, sc = r * r * ci, r * ci, ci, 1
    for _ in range(2):
        check(lib.cdae_conv3x3_fwd(ptr(x), sn, sy, sx, sc, ptr(w), None, None, ptr(y), co, 0, B, r, r, ci, co, 1, 0, ptr(ws), SPLITK_BYTES, stream()))
        check(lib.cdae_conv3x3_dgrad(ptr(dy), co, ptr(w), ptr(dx), ci, B, r, r, ci, co, 1, 0, 0, ptr(ws), SPLITK_BYTES, stream()))
        check(lib.cdae_conv3x3_wgrad(ptr(x), sn, sy, sx, sc, ptr(dy), co, ptr(dw), None, B, r, r, ci, co, 1, 0, 0, ptr(ws), SPLITK_BYTES, stream()))
    torch.cuda.synchronize()
