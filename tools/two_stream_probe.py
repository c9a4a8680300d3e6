"""Dev: which use of a second HIP stream makes the ROCm runtime's helper thread spin?  Per-thread CPU seconds per second of wall
time for a loop of small kernels (a) on one stream, (b) alternating over two streams with no dependencies, (c) with wait_stream both ways,
(d) with one dependency per 16 launches.    python3 tools/two_stream_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
x = torch.zeros(1 << 22, device=dev)
y = torch.zeros(1 << 22, device=dev)
s2 = torch.cuda.Stream()
main = torch.cuda.current_stream()


def run(mode, n=20000):
    torch.cuda.synchronize()
    t0, w0 = bench.thread_cpu(), time.perf_counter()
    for i in range(n):
        x.add_(1.0)
        if mode == "one":
            y.add_(1.0)
        else:
            if mode == "dep" or (mode == "dep16" and i % 16 == 0):
                s2.wait_stream(main)
            with torch.cuda.stream(s2):
                y.add_(1.0)
            if mode == "dep" or (mode == "dep16" and i % 16 == 0):
                main.wait_stream(s2)
        if i % 256 == 255:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    w = time.perf_counter() - w0
    t1 = bench.thread_cpu()
    per = sorted(((name, (cpu - t0.get(tid, ("", 0.0))[1]) / w) for tid, (name, cpu) in t1.items()), key=lambda kv: -kv[1])
    print(f"{mode:6s} wall {w:.2f} s  threads (cores busy):", [(n_, round(v, 2)) for n_, v in per if v > 0.02])


for m in ("one", "two", "dep", "dep16", "one"):
    run(m)
