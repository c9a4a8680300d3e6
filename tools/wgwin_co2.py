"""Dev tool: the one-plane window weight gradient on 64- and on 128-output-channel block tiles (tune key wgwin_co2), single launches and grouped launches of a level."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd._lib import WgItem, check, lib, ptr, stream, splitk_ws, SPLITK_BYTES, precision_scope, tune_scope
dev = torch.device("cuda:0")
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
ws = splitk_ws(dev)
for mode in ("mixed16", "f16x3"):
  with precision_scope(mode):
    for (N, S, Cin, Cout, members) in ((256, 32, 128, 128, 1), (256, 16, 256, 256, 1), (256, 8, 256, 256, 1), (32, 64, 128, 128, 1), (32, 32, 256, 256, 1), (32, 16, 384, 384, 1), (32, 8, 512, 512, 1),
                                       (256, 32, 128, 128, 7), (256, 16, 256, 256, 7), (256, 8, 256, 256, 11), (32, 64, 128, 128, 7), (32, 32, 256, 256, 7), (32, 16, 384, 384, 7), (32, 8, 512, 512, 12)):
        if mode == 'f16x3' and N == 256: continue
        keep, items = [], []
        for m in range(members):
            ap = torch.randn(1, N, S, S, Cin, device=dev).bfloat16(); dp = (torch.randn(1, N, S, S, Cout, device=dev) * 1e-3).bfloat16()
            dw = torch.zeros(Cout, 3, 3, Cin, device=dev); db = torch.zeros(Cout, device=dev)
            keep.append((ap, dp, dw, db))
            items.append(WgItem(ptr(ap[0]), ptr(ap[0]), ptr(dp[0]), ptr(dp[0]), ptr(dw), ptr(db), N, S, S, Cin, Cout, 0))
        arr = (WgItem * members)(*items)
        row = []
        for co2 in (0, 1):
            with tune_scope(wgwin_co2=2 * co2):
                us = t(lambda: check(lib.cdae_conv3x3_wgrad_win_group(arr, members, ptr(ws), SPLITK_BYTES, stream())))
            row.append("co2=%d %7.1f us %6.1f TF" % (co2, us, 2.0 * members * N * S * S * 9 * Cin * Cout / us * 1e-6))
        print(mode, "x%-2d N=%3d %2dx%-2d %4d->%-4d | " % (members, N, S, S, Cin, Cout) + " | ".join(row), flush=True)
