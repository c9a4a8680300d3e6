"""Dev tool: the window weight gradient (wgwin_kernel) by prefetch distance and LDS layout (cdae_tune_set keys wgwin_dist / wgwin_swz), hip-event
timing of single launches on the shapes of config [1] (M32, batch 256, one plane) and of the C64 step (batch 32, two planes / one plane)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd._lib import check, lib, ptr, stream, splitk_ws, SPLITK_BYTES, precision_scope, tune_scope

dev = torch.device("cuda:0")
SHAPES = {
    "mixed16": [(256, 32, 128, 128), (256, 16, 256, 256), (256, 16, 384, 256), (256, 8, 256, 256), (32, 64, 128, 128), (32, 32, 256, 256), (32, 16, 384, 384), (32, 8, 512, 512)],
    "f16x3": [(32, 64, 128, 128), (32, 64, 256, 128), (32, 32, 256, 256), (32, 16, 384, 384), (32, 8, 512, 512)],
}
for mode, shapes in SHAPES.items():
    for (N, S, Cin, Cout) in shapes:
        ap = torch.randn(2, N, S, S, Cin, device=dev).bfloat16()
        dp = (torch.randn(2, N, S, S, Cout, device=dev) * 1e-3).bfloat16()
        dw = torch.zeros(Cout, 3, 3, Cin, device=dev)
        db = torch.zeros(Cout, device=dev)
        ws = splitk_ws(dev)
        line = []
        with precision_scope(mode):
            for dist, swz in ((1, 0), (1, 1), (2, 0), (2, 1)):
                with tune_scope(wgwin_dist=dist, wgwin_swz=swz):
                    def run():
                        check(lib.cdae_conv3x3_wgrad_win(ptr(ap[0]), ptr(ap[1]), ptr(dp[0]), ptr(dp[1]), ptr(dw), ptr(db), N, S, S, Cin, Cout, 0, ptr(ws), SPLITK_BYTES, stream()))
                    for _ in range(3): run()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    reps = 10
                    e0.record()
                    for _ in range(reps): run()
                    e1.record(); torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / reps
                    tf = 2.0 * N * S * S * 9 * Cin * Cout / us * 1e-6
                    line.append("d%d s%d %7.1f us %6.1f TF" % (dist, swz, us, tf))
        print("%-8s N=%3d %2dx%-2d %4d->%-4d | " % (mode, N, S, S, Cin, Cout) + " | ".join(line), flush=True)
