"""Dev: host-side profile (cProfile) of the training step — where the launch thread spends its time."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
bench.train_bench(dev, 1, 0, 2, 2, 32)
pr = cProfile.Profile()
pr.enable()
r = bench.train_bench(dev, 1, 0, 6, 1, 32)
pr.disable()
print(r["ms_per_step"])
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
