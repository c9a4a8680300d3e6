"""Dev: where the host threads of a training step are (statistical: a sampler thread reads sys._current_frames() every millisecond while
bench.train_bench runs) — innermost Python frame per thread, share of samples."""
import collections, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
hist = collections.defaultdict(collections.Counter)
stop = False
def sampler():
    me = threading.get_ident()
    while not stop:
        for tid, fr in sys._current_frames().items():
            if tid == me:
                continue
            f = fr
            key = f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno} {f.f_code.co_name}"
            up = f.f_back
            if up is not None:
                key += f"  <- {os.path.basename(up.f_code.co_filename)}:{up.f_lineno} {up.f_code.co_name}"
            hist[tid][key] += 1
        time.sleep(0.001)
th = threading.Thread(target=sampler, daemon=True)
r0 = bench.train_bench(dev, 1, 0, 10, 3, 32)
th.start()
r = bench.train_bench(dev, 1, 0, int(os.environ.get("STEPS", "60")), 3, 32)
stop = True
th.join()
print("ms/step", round(r["ms_per_step"], 2), "host", round(r["host_cpu_ms_per_step"], 1), r["host_cpu_ms_per_step_by_thread"])
for tid, c in hist.items():
    tot = sum(c.values())
    print(f"-- thread {tid}{' (main)' if tid == threading.main_thread().ident else ''}: {tot} samples")
    for k, v in c.most_common(14):
        print(f"   {100 * v / tot:5.1f}%  {k}")
