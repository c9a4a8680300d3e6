"""Dev tool: one single-plane / two-plane window-wgrad launch under a -DWG_ABL=512 build (tools/wgwin_abl_build.sh 512): the kernel prints cycle stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd._lib import check, lib, ptr, stream, splitk_ws, SPLITK_BYTES, precision_scope
dev = torch.device("cuda:0")
for mode, N, S, Cin, Cout in (("mixed16", 256, 32, 128, 128), ("f16x3", 32, 64, 128, 128)):
    ap = torch.randn(2, N, S, S, Cin, device=dev).bfloat16(); dp = (torch.randn(2, N, S, S, Cout, device=dev) * 1e-3).bfloat16()
    dw = torch.zeros(Cout, 3, 3, Cin, device=dev); db = torch.zeros(Cout, device=dev); ws = splitk_ws(dev)
    print("==", mode, N, S, Cin, Cout, flush=True)
    with precision_scope(mode):
        for _ in range(2):
            check(lib.cdae_conv3x3_wgrad_win(ptr(ap[0]), ptr(ap[1]), ptr(dp[0]), ptr(dp[1]), ptr(dw), ptr(db), N, S, S, Cin, Cout, 0, ptr(ws), SPLITK_BYTES, stream()))
            torch.cuda.synchronize()
