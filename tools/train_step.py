"""Dev tool: a few training steps (C64, batch 32) for rocprofv3 kernel traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
r = bench.train_bench(dev, 1, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 3, 2, int(sys.argv[2]) if len(sys.argv) > 2 else 32)
print(r)
