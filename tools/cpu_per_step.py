"""Dev: host CPU seconds per training step (all threads) next to the wall time — what 8 ranks on one host would need."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
bench.train_bench(dev, 1, 0, 2, 2, 32)
c0, w0 = time.process_time(), time.perf_counter()
r = bench.train_bench(dev, 1, 0, 20, 1, 32)
c1, w1 = time.process_time(), time.perf_counter()
print("ms/step", r["ms_per_step"], "whole call: cpu s", c1 - c0, "wall s", w1 - w0, "threads", torch.get_num_threads(), "cpus", len(os.sched_getaffinity(0)))
