"""Dev tool: hip-event timing of single window-wgrad launches (default tune) on a few shapes — the AB_CMD of tools/ab_lib.sh for wgwin ablation builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd._lib import check, lib, ptr, stream, splitk_ws, SPLITK_BYTES, precision_scope
dev = torch.device("cuda:0")
out = []
for mode, N, S, Cin, Cout in (("mixed16", 256, 32, 128, 128), ("mixed16", 256, 16, 256, 256), ("mixed16", 32, 64, 128, 128), ("f16x3", 32, 64, 128, 128), ("f16x3", 32, 32, 256, 256)):
    ap = torch.randn(2, N, S, S, Cin, device=dev).bfloat16()
    dp = (torch.randn(2, N, S, S, Cout, device=dev) * 1e-3).bfloat16()
    dw = torch.zeros(Cout, 3, 3, Cin, device=dev); db = torch.zeros(Cout, device=dev); ws = splitk_ws(dev)
    with precision_scope(mode):
        def run():
            check(lib.cdae_conv3x3_wgrad_win(ptr(ap[0]), ptr(ap[1]), ptr(dp[0]), ptr(dp[1]), ptr(dw), ptr(db), N, S, S, Cin, Cout, 0, ptr(ws), SPLITK_BYTES, stream()))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        out.append("%s %dx%d %d->%d n%d: %.1f us" % (mode, S, S, Cin, Cout, N, e0.elapsed_time(e1) * 100))
print("value " + " | ".join(out))
