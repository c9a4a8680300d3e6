"""Dev: host-side profile (cProfile) of training steps, backward included (autograd multithreading off so that the backward
functions run on the profiled thread)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from improved_diffusion import script_util as su
from improved_diffusion.image_datasets import load_data
from improved_diffusion.train_util import TrainLoop
import bench
dev = torch.device("cuda:0")
cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True}
model, diff = su.create_model_and_diffusion(**cfg)
bench.randomize(model, 4321)
model.to(dev).train()
data = load_data(data_dir="synthetic", batch_size=32, image_size=64, in_channels=3, n_vars=4, seed=0, device=dev)
loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=32, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                 save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3,
                 use_fp16=os.environ.get("FP16") == "1")          # FP16=1: the 16-bit torso
diff.kl_weight = 0.1
def steps(n):
    for _ in range(n):
        b, c = next(data); loop.forward_backward(b, c); loop.optimize_normal()
    torch.cuda.synchronize()
steps(3)
torch.autograd.set_multithreading_enabled(False)
steps(2)
pr = cProfile.Profile()
t0 = time.perf_counter(); c0 = time.process_time()
pr.enable(); steps(10); pr.disable()
print("wall ms/step", 100 * (time.perf_counter() - t0), "cpu ms/step", 100 * (time.process_time() - c0))
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(30)
