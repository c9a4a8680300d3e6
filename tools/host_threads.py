"""Dev: CPU time per thread of the C64 batch-32 training step (30 steps), for A/B runs under runtime environment variables."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
r = bench.train_bench(torch.device("cuda:0"), 1, 0, 30, 5, 32)
print("ms/step", round(r["ms_per_step"], 2), "cpu ms/step", round(r["host_cpu_ms_per_step"], 1), r["host_cpu_ms_per_step_by_thread"])
