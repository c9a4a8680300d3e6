#!/usr/bin/env python3
"""bench.py — headline benchmark of the CausalDiffAE diffusion hot path on MI355X.

A "step" = one DDIM denoise step (UNet forward + fused DDIM update) over one batch of synthetic 64x64
images: BASELINE.json config "Pendulum 64x64, 4 causal vars, DDIM-100 counterfactual sampling", per-GPU
batch 128 (the per-GPU share of config 5).  The sampling batch is sharded over ranks with no collective in
the loop (weak scaling).  Prints ONE JSON line (rank 0).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Headline: EXACTLY K timed steps per timed region (graph replay of the step the public ddim_sample_loop captures), `--regions` regions
back to back (default 3): `value` is the median region, min / max beside it.  Secondary legs, all in the same line (single-GPU runs
only unless stated): the whole public DDIM-100 loop (default call and use_graph=False), the guided loop (w = 2: two forwards per
step), IEEE-fp32 products (sampling and training), the C64 training step (every rank), BASELINE config [1] (M32, batch 256: f16x3 and
the mixed16 torso), the CPU oracle baseline.
"""
import argparse
import json
import os
import statistics
import sys
import time

os.environ.setdefault("DEBUG_CLR_MAX_BATCH_SIZE", "32768")      # causaldiffae_amd/_lib.py sets the same default; here before torch.cuda.is_available() initialises HIP

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate
F16_MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (spec)
GFLOP_PER_IMAGE_STEP_P64 = 60.63       # SURVEY §8d (torch FlopCounter on the reference forward)
GFLOP_PER_IMAGE_TRAIN_C64 = 181.86
GFLOP_PER_IMAGE_TRAIN_M32 = 3 * 12.63


def randomize(model, seed):
    """Random (non-zero) weights of the architecture: zero-filled operands would flatter the clocks."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() > 1:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / fan_in) ** 0.5)
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
        for name, b in model.named_buffers():
            if name.endswith("running_var"):
                b.copy_(1.0 + 0.1 * torch.rand(b.shape, generator=g))


def host_cores():
    """CPUs this process may actually use: min(affinity, cgroup quota) — the GPU box exposes 256 logical CPUs
    under a 16-CPU quota, and oversubscribing torch's thread pool there is 10x slower."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def thread_cpu():
    """CPU seconds of every thread of this process (utime + stime from /proc/self/task): who spends the host cores of a step"""
    tick, out = os.sysconf("SC_CLK_TCK"), {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (f[f.index("(") + 1:f.rindex(")")], (int(rest[11]) + int(rest[12])) / tick)
        except (OSError, ValueError):
            pass
    return out


def spread(values):
    """median / min / max of per-region rates (the pool's boxes differ by +-3 %, one short sample is not a stable number)"""
    return {"median": statistics.median(values), "min": min(values), "max": max(values), "regions": len(values)}


def _cpu_train_step(U, D, sd, cfg, sch, x0, c, y, steps):
    """Oracle training steps on the host: training_losses + backward + AdamW/EMA (reference train_util.py:232-297)."""
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k and "num_batches" not in k]
    params = [sd[k].clone().requires_grad_(True) for k in names]
    live = dict(sd)
    live.update(dict(zip(names, params)))
    m, v = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    ema = [p.detach().clone() for p in params]
    N = x0.shape[0]
    g = torch.Generator().manual_seed(7)
    t0 = None
    for step in range(steps + 1):
        if step == 1:
            t0 = time.perf_counter()
        t = torch.randint(0, 1000, (N,), generator=g)
        noise, eps_z = torch.randn(x0.shape, generator=g), torch.randn(N, 512, generator=g)
        fn = lambda x_t, tm, xs: U.unet_forward(live, cfg, x_t, tm, y=y, x_start=xs, eps_z=eps_z, training=True, new_stats={})
        terms = D.training_losses(sch, fn, x0, t, noise, c=c, rep_cond=True, causal_modeling=True, kl_weight=0.1)
        terms["loss"].mean().backward()
        with torch.no_grad():
            D.adamw_ema_step([p.data for p in params], [p.grad for p in params], m, v, ema, step + 1)
            for p in params:
                p.grad = None
    return steps / (time.perf_counter() - t0)


def cpu_baseline(ddim_steps=6, batch=16, train_steps=3):
    """The oracle (torch-CPU restatement of the reference, pinned to the reference's own outputs by tests/test_oracle_golden.py)
    timed on this host's cores: the headline P64 DDIM step, the C64 training step, and BASELINE config [0] (MorphoMNIST 32x32,
    T = 1000, batch 16: sampling step and training step).  Bounded samples, a few tens of seconds in all."""
    from oracle import diffusion_ref as D
    from oracle import unet_ref as U
    from oracle.closed_form import fill_state_dict, synth
    cores = host_cores()
    torch.set_num_threads(cores)
    note = f"{cores} threads (= cgroup CPU quota; host shows {os.cpu_count()} logical CPUs)"
    # --- headline: P64 DDIM-100 step
    cfg = U.default_cfg(image_size=64, in_channels=4, n_vars=4, rep_cond=True, causal_modeling=True)
    sd = fill_state_dict(U.param_spec(cfg))
    sch = D.Schedule(1000, "linear", "ddim100", True)
    x = synth("bench.cpu.x", (batch, 4, 64, 64))
    z = synth("bench.cpu.z", (batch, 512))
    fn = lambda xx, tm: U.unet_forward(sd, cfg, xx, tm, z=z)[0]
    with torch.no_grad():
        D.sample_loop(sch, fn, x, ddim=True, n_steps=1)
        t0 = time.perf_counter()
        D.sample_loop(sch, fn, x, ddim=True, n_steps=ddim_steps)
        dt = time.perf_counter() - t0
    out = {"value": batch * ddim_steps / dt, "unit": "image-steps/s", "cores": cores, "kind": "port",
           "sample": f"oracle (torch-CPU restatement), P64 DDIM step, batch {batch}, {ddim_steps} steps after 1 warm-up, {note}"}
    # --- C64 training step (BASELINE config [3]'s model, batch 16 on the host)
    cfg_c = U.default_cfg(image_size=64, in_channels=3, n_vars=4, rep_cond=True, causal_modeling=True)
    sd_c = fill_state_dict(U.param_spec(cfg_c))
    sps = _cpu_train_step(U, D, sd_c, cfg_c, D.Schedule(1000, "linear", "", True), synth("bench.cpu.x0c", (batch, 3, 64, 64), 0.0, 1.0),
                          synth("bench.cpu.cc", (batch, 4), 0.0, 1.0), None, train_steps)
    out["train"] = {"value": sps, "unit": "train-steps/s", "images_per_sec": sps * batch, "cores": cores, "kind": "port",
                    "sample": f"oracle training_losses + backward + AdamW/EMA, C64 batch {batch}, {train_steps} steps after 1 warm-up, {note}"}
    # --- BASELINE config [0]: MorphoMNIST 32x32, 2 causal vars, T = 1000, batch 16
    cfg_m = U.default_cfg(image_size=32, in_channels=1, n_vars=2, rep_cond=True, causal_modeling=True, class_cond=True)
    sd_m = fill_state_dict(U.param_spec(cfg_m))
    sch_m = D.Schedule(1000, "linear", "", True)
    xm, zm = synth("bench.cpu.xm", (16, 1, 32, 32)), synth("bench.cpu.zm", (16, 512))
    ym = torch.arange(16) % 10
    fm = lambda xx, tm: U.unet_forward(sd_m, cfg_m, xx, tm, y=ym, z=zm)[0]
    with torch.no_grad():
        D.sample_loop(sch_m, fm, xm, ddim=False, n_steps=1, noises=[torch.zeros_like(xm)] * 1000)
        t0 = time.perf_counter()
        D.sample_loop(sch_m, fm, xm, ddim=False, n_steps=10, noises=[torch.zeros_like(xm)] * 1000)
        dtm = time.perf_counter() - t0
    sps_m = _cpu_train_step(U, D, sd_m, cfg_m, sch_m, synth("bench.cpu.x0m", (16, 1, 32, 32), 0.0, 1.0), synth("bench.cpu.cm", (16, 2), 0.0, 1.0), ym, 3)
    out["config0_m32"] = {"p_sample_image_steps_per_sec": 16 * 10 / dtm, "samples_per_sec_T1000": 16 * 10 / dtm / 1000.0,
                          "train_steps_per_sec": sps_m, "cores": cores, "kind": "port",
                          "sample": f"oracle, MorphoMNIST 32x32 C=1, 2 causal vars, class_cond, T=1000, batch 16: 10 p_sample steps and 3 training "
                                    f"steps after 1 warm-up each, {note}"}
    return out


def train_bench(dev, world, rank, steps, warmup, batch, regions=1, image_size=64, in_channels=3, n_vars=4, use_fp16=False, class_cond=False,
                workload=None):
    """Training leg of the metric: CausalCircuit 64x64 C=3 (BASELINE config 4) by default, per-GPU batch `batch`, one optimizer
    step = forward + backward + bucketed gradient all-reduce (RCCL, overlapped with backward) + fused AdamW/EMA.  `regions` timed
    regions of exactly `steps` steps each; the reported value is the median region (max over ranks per region)."""
    import numpy as np
    import causaldiffae_amd
    from improved_diffusion import script_util as su
    from improved_diffusion.image_datasets import load_data
    from improved_diffusion.train_util import TrainLoop
    cfg = {**su.model_and_diffusion_defaults(), "image_size": image_size, "in_channels": in_channels, "n_vars": n_vars, "rep_cond": True,
           "causal_modeling": True, "class_cond": class_cond}
    model, diff = su.create_model_and_diffusion(**cfg)
    randomize(model, 4321)
    model.to(dev).train()
    np.random.seed(1000 + rank)
    # HBM-resident synthetic pool + gather kernel: the feed the training script uses (`--host_feed` off), no host work per step
    data = load_data(data_dir="synthetic", batch_size=batch, image_size=image_size, in_channels=in_channels, n_vars=n_vars, seed=rank,
                     class_cond=class_cond, device=None if os.environ.get("CDAE_BENCH_HOST_FEED") == "1" else dev)
    loop = TrainLoop(model=model, diffusion=diff, data=data, batch_size=batch, microbatch=-1, lr=1e-4, ema_rate="0.9999",
                     log_interval=10 ** 9, save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=n_vars,
                     causal_modeling=True, in_channels=in_channels, use_fp16=use_fp16)
    diff.kl_weight = 0.1
    nparams = sum(p.numel() for p in model.parameters())

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(1, warmup)):
        b, c = next(data)
        loop.forward_backward(b, c)
        loop.optimize_normal()
    rates, cpus = [], []
    th0 = thread_cpu()
    for _ in range(regions):
        sync()
        t0, c0 = time.perf_counter(), time.process_time()
        for _ in range(steps):
            b, c = next(data)
            loop.forward_backward(b, c)
            loop.optimize_normal()
        sync()
        dt, cpu = time.perf_counter() - t0, time.process_time() - c0
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt, cpu], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt, cpu = tt[0].item(), tt[1].item()          # the slowest rank's wall time, the hungriest rank's CPU time
        rates.append(steps / dt)
        cpus.append(1e3 * cpu / steps)
    th1 = thread_cpu()
    # what each of EIGHT ranks under this box's quota runs (ops._host_cores_per_rank() < 2.5: weight gradients on the launch stream, no second
    # stream): the same loop with the side stream off, a short region — its CPU per step is the number the eight-rank budget is judged on
    w8 = None
    from causaldiffae_amd import ops as _ops
    if world == 1 and _ops.wgrad_side_stream_on():
        _ops.side_join()
        _ops._WGRAD_SIDE_ON = False
        try:
            for _ in range(5):             # (the other policy's tensor lifetimes: let the caching allocator settle first)
                b, c = next(data)
                loop.forward_backward(b, c)
                loop.optimize_normal()
            k8, runs = max(10, min(steps, 20)), []
            for _ in range(3):
                sync()
                t0, c0 = time.perf_counter(), time.process_time()
                for _ in range(k8):
                    b, c = next(data)
                    loop.forward_backward(b, c)
                    loop.optimize_normal()
                sync()
                runs.append(((time.perf_counter() - t0) / k8, (time.process_time() - c0) / k8))
            dt8, cpu8 = sorted(runs)[1]        # the median region
            w8 = {"value": 1.0 / dt8, "ms_per_step": 1e3 * dt8, "host_cpu_ms_per_step": 1e3 * cpu8, "host_cpu_over_step": cpu8 / dt8, "wgrad_side_stream": False,
                  "steps": k8, "regions": 3}
        finally:
            _ops._WGRAD_SIDE_ON = True
    # kernel-family milliseconds of one step (HIP events around every launch of the library, outside the timed regions): where the step goes
    from causaldiffae_amd import _lib as _l
    torch.cuda.synchronize()
    _l.prof_enable(True)
    _l.prof_read()
    for _ in range(2):
        b, c = next(data)
        loop.forward_backward(b, c)
        loop.optimize_normal()
    fam = _l.prof_read()
    _l.prof_enable(False)
    family_ms = {k: round(v["ms"] / 2, 3) for k, v in fam.items()}
    per_thread = sorted(((name, 1e3 * (cpu - th0.get(tid, ("", 0.0))[1]) / (steps * regions)) for tid, (name, cpu) in th1.items()), key=lambda kv: -kv[1])
    loss = float(loop.last_losses["loss"].mean().item())
    sps = statistics.median(rates)
    gflop = GFLOP_PER_IMAGE_TRAIN_C64 if image_size == 64 else GFLOP_PER_IMAGE_TRAIN_M32
    prec = getattr(model, "_cdae_precision", None) or causaldiffae_amd.get_precision()
    dtype = {"f16x3": "f32 in/out/accumulate/optimizer; forward products f16x3 (2^-22), bf16x3 (2^-16) products in dgrad/wgrad",
             "fp32": "f32 everywhere: IEEE fp32 products on v_mfma_f32_32x32x2_f32, forward and backward",
             "mixed16": "f32 master weights / accumulate / optimizer; single f16 plane forward, single bf16 plane backward (the reduced-precision torso)"}[prec]
    roof = {"f16x3": F16_MFMA_PEAK_TFLOPS / 3, "fp32": FP32_MFMA_PEAK_TFLOPS, "mixed16": F16_MFMA_PEAK_TFLOPS}[prec]
    tf = batch * world * sps * gflop / 1e3
    host_ms = statistics.median(cpus)
    quota = host_cores()
    return {"value": sps, "unit": "train-steps/s", "ms_per_step": 1e3 / sps, "spread": spread(rates), "batch_per_gpu": batch,
            "global_batch": batch * world, "images_per_sec": batch * world * sps, "model_tflops": tf, "roof_tflops": roof * world,
            "frac_of_roof": tf / (roof * world), "precision_mode": prec, "dtype": dtype,
            "steps": steps, "warmup": warmup, "last_loss": loss,
            # (sum of kernel durations by family; with the side stream on, concurrent kernels both count in full)
            "family_ms_per_step": family_ms,
            # CPU time of the hungriest rank (all its threads) per step; `world` such ranks share the cgroup quota on one node
            "host_cpu_ms_per_step": host_ms, "host_cpu_over_step": host_ms * sps / 1e3, "host_cpu_quota_cores": quota,
            "host_bound_risk": bool(world * host_ms * sps / 1e3 > 0.8 * quota),
            # the driver's scaling run puts EIGHT such ranks under one cgroup quota: cores they would need vs 0.8 x the quota (this process's
            # quota stands in for the node's: the 1-GPU lease and the 8-GPU node are provisioned alike)
            # (from the eight-rank policy's own measurement where this run made one: `world8_policy`)
            "host_cores_needed_at_world8": 8 * (w8["host_cpu_over_step"] if w8 else host_ms * sps / 1e3),
            "host_bound_risk_at_world8": bool(8 * (w8["host_cpu_over_step"] if w8 else host_ms * sps / 1e3) > 0.8 * quota),
            "world8_policy": w8,
            # weight gradients on a second HIP stream (ops.side_launch): on where a rank has >= 2.5 host cores to itself (the stream keeps a
            # runtime helper thread busy), i.e. OFF for eight ranks under a 16-core quota — see `world8_policy` for that configuration's numbers
            "wgrad_side_stream": bool(__import__("causaldiffae_amd").ops.wgrad_side_stream_on()),
            "host_cpu_ms_per_step_by_thread": [[n, round(v, 2)] for n, v in per_thread[:6] if v >= 0.05],
            "dist_backend": (torch.distributed.get_backend() if world > 1 else None),
            "workload": workload or f"CausalCircuit 64x64 C=3 CausalDiffAE training step (fwd+bwd+all-reduce+AdamW/EMA), {nparams / 1e6:.1f}M params"}


def self_launch(n):
    """`python bench.py --gpus N` started bare (no launcher, WORLD_SIZE unset): start N fresh rank processes of this same command — one
    per GPU, env:// rendezvous on 127.0.0.1 — BEFORE this process touches the GPU, relay rank 0's JSON line, exit non-zero if any rank
    fails.  (Never a re-exec of a process that initialised HIP; torch.cuda.device_count() does not.)  With fewer devices than ranks
    (a one-GPU test box) the ranks share devices and gloo stands in for RCCL unless CDAE_DIST_BACKEND says otherwise."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ndev = torch.cuda.device_count()
    procs = []
    for r in range(n):
        env = {**os.environ, "RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
               "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
        if 0 < ndev < n:
            env.setdefault("CDAE_DIST_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True if r == 0 else None))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while any(p.poll() is None for p in procs):
        failed = next((p for p in procs if p.poll() not in (None, 0)), None)
        if failed is not None:            # a dead rank leaves its peers waiting in the next collective: end them (exact PIDs we started)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(10)
    sys.stdout.write("".join(c for c in chunks if c))
    sys.stdout.flush()
    if failed is not None or any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        sys.exit((failed.returncode if failed is not None else next(c for c in codes if c)) or 1)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=3, help="timed regions of exactly --steps steps each (median reported)")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--precision", choices=["f16x3", "fp32"], default=None, help="arithmetic of the K-contiguous contractions")
    ap.add_argument("--train-batch", type=int, default=32)
    ap.add_argument("--train-steps", type=int, default=50)
    ap.add_argument("--no-fp32", action="store_true", help="skip the secondary IEEE-fp32-product legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the public-loop, guidance and config [1] legs")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)            # does not return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CDAE_WATCHDOG_S"):          # dev: every thread's stack to stderr and exit if the run is still going after that many seconds
        import ctypes, faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CDAE_WATCHDOG_S"]), exit=True)
        ctypes.CDLL(None).prctl(0x59616D61, ctypes.c_ulong(-1), 0, 0, 0)      # PR_SET_PTRACER_ANY: a debugger started beside the job may attach
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product path has no CPU fallback)"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # "nccl" is RCCL on ROCm; CDAE_DIST_BACKEND=gloo exists only to exercise the N>1 code path on a 1-GPU box
        import datetime
        tmo = {"timeout": datetime.timedelta(seconds=int(os.environ["CDAE_DIST_TIMEOUT_S"]))} if os.environ.get("CDAE_DIST_TIMEOUT_S") else {}
        dist.init_process_group(os.environ.get("CDAE_DIST_BACKEND", "nccl"), init_method="env://", **tmo)

    import causaldiffae_amd  # noqa: F401
    from causaldiffae_amd import _lib
    from causaldiffae_amd.gaussian_diffusion import _GraphStep
    from improved_diffusion import script_util as su
    from improved_diffusion.nn import reparameterize
    from improved_diffusion.unet import ADJACENCY

    if args.precision:
        causaldiffae_amd.set_precision(args.precision)
    prec0 = causaldiffae_amd.get_precision()
    # what THIS box sustains (register-resident MFMA loop, flat HBM copy; outside every timed region): the pool's boxes differ by +-3 %,
    # the fractions below are reported against the nominal peaks AND against these
    box = _lib.calibrate(dev) if rank == 0 else None
    single = world == 1
    # ---- training legs FIRST, in a process that has not replayed a HIP graph yet: the sampling legs below leave runtime helper threads
    # behind (one of them busy-waiting: 6-28 ms of CPU per training step measured when the order is reversed) that a training job does not have
    train = None
    if not args.no_train:
        try:
            train = train_bench(dev, world, rank, args.train_steps, 5, args.train_batch, regions=max(1, args.regions))
            if single and not args.no_fp32:
                causaldiffae_amd.set_precision("fp32")
                try:
                    torch.cuda.empty_cache()
                    train["fp32_mode"] = train_bench(dev, world, rank, 6, 3, args.train_batch, regions=3)      # (median of three regions: a single region of this leg has shown 66 and 73 ms on one box)
                finally:
                    causaldiffae_amd.set_precision(prec0)
            if single and not args.no_extra:
                # BASELINE config [1]: MorphoMNIST 32x32 CausalDiffAE training, batch 256 — parity mode and the reduced-precision torso
                torch.cuda.empty_cache()
                m32 = dict(image_size=32, in_channels=1, n_vars=2, class_cond=True)
                wl = "BASELINE config [1]: MorphoMNIST 32x32 C=1, 2 causal vars, class-conditional, batch 256 training step"
                train["config1_m32_b256"] = {"f16x3": train_bench(dev, world, rank, 20, 3, 256, regions=2, workload=wl, **m32)}
                torch.cuda.empty_cache()
                train["config1_m32_b256"]["mixed16"] = train_bench(dev, world, rank, 20, 3, 256, regions=2, use_fp16=True, workload=wl + " (use_fp16)", **m32)
                a, b = train["config1_m32_b256"]["mixed16"]["value"], train["config1_m32_b256"]["f16x3"]["value"]
                train["config1_m32_b256"]["mixed16_over_f16x3"] = a / b
                # the same torso under the C64 leg's own workload (reference: `--use_fp16 True` on the 64 x 64 configuration); the leg's headline
                # value above stays the parity mode's
                torch.cuda.empty_cache()
                train["mixed16_torso"] = train_bench(dev, world, rank, 40, 5, args.train_batch, regions=3, use_fp16=True)      # (2 x 20 steps read 17.1 .. 18.7 ms on the same code)
                train["mixed16_torso_over_f16x3"] = train["mixed16_torso"]["value"] / train["value"]
        except Exception as e:                      # never lose the headline line to the secondary leg
            train = {**(train or {}), "error": f"{type(e).__name__}: {e}"[:300]}
    cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 4, "n_vars": 4, "rep_cond": True,
           "causal_modeling": True, "timestep_respacing": "ddim100"}
    model, diff = su.create_model_and_diffusion(**cfg)
    randomize(model, 1234)
    model.to(dev).eval()
    N = args.batch
    g = torch.Generator().manual_seed(100 + rank)          # each rank samples its own shard of the global batch
    x0 = torch.rand(N, 4, 64, 64, generator=g).to(dev)
    with torch.no_grad():
        # counterfactual pattern of scripts/image_causaldae_test.py:535-594 (pendulum branch)
        A = torch.tensor(ADJACENCY["pendulum"], dtype=torch.float32)
        mu, var = model.rep_emb.encode(x0)
        z_post = model.causal_mask.nonlinearity_add_back_noise(mu, model.causal_mask.causal_masking(mu, A))
        z_post[:, :128] = 0.2
        z = reparameterize(z_post, torch.full_like(mu, 0.001), eps=torch.randn(N, 512, generator=g).to(dev))
        t_last = torch.full((N,), diff.num_timesteps - 1, dtype=torch.int64, device=dev)
        x_t = diff.q_sample(x0, t_last, noise=torch.randn(N, 4, 64, 64, generator=g).to(dev))
        kw = dict(z=z)

        T = diff.num_timesteps
        steps_tab = diff._step_table(dev, N)

        def sync():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()

        def max_over_ranks(dt):
            if world > 1:
                tt = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = tt.item()
            return dt

        def timed_ddim(steps, warmup, regions=1, w=None):
            """`regions` regions of EXACTLY `steps` timed DDIM steps (the graph the public loop replays, unless --no-graph) after `warmup`
            untimed ones; seconds per region, max over ranks"""
            if args.no_graph:
                state = {"img": x_t.clone()}

                def do_step(k):
                    state["img"] = diff.ddim_sample(model, state["img"], steps_tab[k % T], model_kwargs=kw, w=w)["sample"]
            else:
                runner = _GraphStep(diff, model, x_t.clone(), kw, True, w)

                def do_step(k):
                    runner.step(k % T)
            for k in range(max(1, warmup)):
                do_step(k)
            out, k0 = [], warmup
            for _ in range(regions):
                sync()
                t0 = time.perf_counter()
                for k in range(steps):
                    do_step(k0 + k)
                sync()
                out.append(max_over_ranks(time.perf_counter() - t0))
                k0 += steps
            return out

        def public_loop(**kwargs):
            """One call of the PUBLIC API exactly as scripts/image_causaldae_test.py:587-594 makes it: all T steps, seconds (max over ranks)"""
            sync()
            t0 = time.perf_counter()
            diff.ddim_sample_loop(model, (N, 4, 64, 64), noise=x_t, model_kwargs=kw, **kwargs)
            sync()
            return max_over_ranks(time.perf_counter() - t0)

        def roofline(prec, eager_steps=2):
            """HIP events on the launch stream around every launch of the contraction families during eager steps, outside the timed
            region.  The dominant kernel (the window conv, convwin_kernel<f16, 9 taps>) is reported by itself; `all_contractions` is every
            contraction family."""
            img2 = x_t.clone()
            diff.ddim_sample(model, img2, steps_tab[0], model_kwargs=kw)
            torch.cuda.synchronize()
            _lib.prof_enable(True)
            for k in range(eager_steps):
                img2 = diff.ddim_sample(model, img2, steps_tab[k], model_kwargs=kw)["sample"]
            prof = _lib.prof_read()
            _lib.prof_enable(False)
            # f16x3: every algorithmic multiply-add is 3 f16 MFMA multiply-adds, so the matrix-core roof for ALGORITHMIC flops is the
            # dense f16 peak / 3; fp32: the fp32 MFMA peak
            peak = F16_MFMA_PEAK_TFLOPS / 3.0 if prec == "f16x3" else FP32_MFMA_PEAK_TFLOPS
            cw, ig = prof["convwin"], prof["igemm"]
            fams = [prof[k] for k in ("igemm", "convwin", "convwin_dgrad", "convwin_up")]
            fam_ms, fam_work, fam_n = (sum(f[k] for f in fams) for k in ("ms", "work", "launches"))
            dom = cw if cw["launches"] > 0 and cw["ms"] >= 0.4 * fam_ms else ig
            name = ("convwin_kernel<f16, 9 taps> (convwin.hip: stride-1 conv3x3 on pre-split f16 hi/lo planes, window resident in LDS, "
                    "v_mfma_f32_16x16x32_f16 x3 per product, fp32 accumulate)" if dom is cw else
                    ("igemm_kernel (v_mfma_f32_32x32x2_f32, IEEE fp32 products)" if prec == "fp32" else "pswin / ps / igemm_kernel family (f16x3)"))
            ach = dom["work"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
            r = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s (algorithmic 2MNK)", "frac": ach / peak,
                 "precision_mode": prec, "launches_per_step": dom["launches"] // eager_steps,
                 "avg_launch_us": 1e3 * dom["ms"] / max(1, dom["launches"]), "flops_per_launch_avg": dom["work"] / max(1, dom["launches"]),
                 "share_of_contraction_time": dom["ms"] / fam_ms if fam_ms > 0 else None,
                 "executed_mfma_tflops": ach * (3.0 if prec == "f16x3" else 1.0),
                 # 2MNK as EXECUTED: the three upsample convs run in their folded sub-pixel form (2.25x fewer multiply-adds than the
                 # reference's formulation), so this is below the reference-algorithm rate `model_tflops` implies
                 "flops_convention": "executed 2MNK per launch (sub-pixel up-convs at folded size)",
                 # against what a register-resident v_mfma_f32_16x16x32_f16 loop sustains on THIS box in THIS process (`box`,
                 # cdae_calib_mfma: ~1900-1980 TFLOP/s at the ~1.85 GHz the chip holds under MFMA load; / 3 in f16x3 terms)
                 "frac_of_sustained_mfma": (ach / (box["mfma_sustained_tflops"] / 3.0)) if prec == "f16x3" and box else None,
                 "all_contractions": {"achieved": fam_work / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0, "launches_per_step": fam_n // eager_steps,
                                      "ms_per_step": fam_ms / eager_steps},
                 "family_ms_per_step": {k: v["ms"] / eager_steps for k, v in prof.items()}}
            r["all_contractions"]["frac"] = r["all_contractions"]["achieved"] / peak
            # HBM traffic of the dominant kernel: PMC counters cannot be read in-process (separate rocprofv3 --pmc passes of this same
            # command, folded by tools/pmc_summary.py and committed under profiles/)
            traffic = traffic_src = None
            # (convwin: the 9-TAP instantiation's own summary — the 4-tap sub-pixel kernel is a separate kernel name and a separate file)
            stem = "convwin9" if dom is cw else "igemm"
            cands = [f"r06_{stem}_pmc_summary_{prec}.json", f"r05_{stem}_pmc_summary_{prec}.json", f"r04_{stem}_pmc_summary_{prec}.json"] + ([f"r03_convwin_pmc_summary_{prec}.json"] if dom is cw else [f"r01_igemm_pmc_summary_{prec}.json"])
            pmc_file = next((os.path.join(ROOT, "profiles", c) for c in cands if os.path.exists(os.path.join(ROOT, "profiles", c))), None)
            if pmc_file and N == 128:
                pm = json.load(open(pmc_file))
                traffic, traffic_src = pm["hbm_traffic_bytes_per_launch"], f"profiles/{os.path.basename(pmc_file)} (rocprofv3 --pmc, same workload)"
            alg = dom["bytes"] / max(1, dom["launches"]) if dom["bytes"] > 0 else None
            r.update({"traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src, "traffic_measured_in_run": False,
                      "algorithmic_bytes_per_launch": alg, "traffic_ratio": (traffic / alg) if traffic and alg else None})
            # the fused sub-pixel up-conv (convwin_kernel<f16, 4 taps>: nearest-2x + conv3x3 as four 2x2 phases of the low-res input) by itself
            up = prof["convwin_up"]
            if up["launches"] > 0 and up["ms"] > 0:
                ach4 = up["work"] / (up["ms"] * 1e-3) / 1e12
                alg4 = up["bytes"] / up["launches"] if up["bytes"] > 0 else None
                pm4 = next((os.path.join(ROOT, "profiles", c) for c in (f"r06_convwin4_pmc_summary_{prec}.json", f"r05_convwin4_pmc_summary_{prec}.json", f"r04_convwin4_pmc_summary_{prec}.json")
                            if os.path.exists(os.path.join(ROOT, "profiles", c))), None)
                t4 = json.load(open(pm4))["hbm_traffic_bytes_per_launch"] if pm4 and N == 128 else None
                r["upconv_4tap"] = {"kernel": "convwin_kernel<f16, 4 taps>", "bound": "mfma", "achieved": ach4, "peak": peak, "frac": ach4 / peak,
                                    "unit": "TFLOP/s (executed 2MNK, folded 2x2 phases)", "launches_per_step": up["launches"] // eager_steps,
                                    "avg_launch_us": 1e3 * up["ms"] / up["launches"], "ms_per_step": up["ms"] / eager_steps,
                                    "frac_of_sustained_mfma": (ach4 / (box["mfma_sustained_tflops"] / 3.0)) if prec == "f16x3" and box else None,
                                    "algorithmic_bytes_per_launch": alg4, "traffic": t4, "traffic_ratio": (t4 / alg4) if t4 and alg4 else None,
                                    "traffic_source": f"profiles/{os.path.basename(pm4)}" if t4 else None, "traffic_measured_in_run": False}
            return r

        region_s = timed_ddim(args.steps, args.warmup, max(1, args.regions))
        dt = statistics.median(region_s)
        roof = roofline(prec0) if rank == 0 else None
        extra = {}
        if single and not args.no_extra:
            # the drop-in call: the whole DDIM-100 loop through the public API (default = graph replay where eligible), and eagerly
            public_loop()                                       # untimed: allocator warm-up + graph capture paths
            d_s = [public_loop() for _ in range(2)]
            e_s = [public_loop(use_graph=False)]
            extra["public_ddim_sample_loop"] = {
                "call": "diffusion.ddim_sample_loop(model, (128, 4, 64, 64), noise=x_T, model_kwargs={'z': z})  [scripts/image_causaldae_test.py:587-594]",
                "steps_per_call": T, "default_ms_per_step": 1e3 * min(d_s) / T, "default_calls_s": d_s,
                "default_image_steps_per_sec": N * T / min(d_s), "default_includes": "graph capture of one step + 100 replays per call",
                "eager_ms_per_step": 1e3 * min(e_s) / T, "eager_image_steps_per_sec": N * T / min(e_s)}
            # config 5 (classifier-free masking / guidance): two forwards per step, gaussian_diffusion.py:277-285 — "report both"
            kg = max(10, min(args.steps, 20))
            g_s = timed_ddim(kg, 2, 2, w=2.0)
            extra["guided_w2"] = {"value": N * kg / min(g_s), "unit": "image-steps/s", "ms_per_step": 1e3 * min(g_s) / kg, "steps": kg, "regions": 2,
                                  "forwards_per_step": 2, "model_tflops": 2 * N * kg / min(g_s) * GFLOP_PER_IMAGE_STEP_P64 / 1e3,
                                  "workload": "the same P64 DDIM step with guidance w = 2 (conditional + unconditional forward per step)"}
            # SURVEY §8d config 3 names N in {16, 128}: the small batch is launch- and latency-bound (the ~300-launch graph matters most there)
            n16 = 16
            x16, kw16 = x_t[:n16].clone(), dict(z=z[:n16].clone())
            k16 = max(20, min(args.steps, 50))

            def run16(graph):
                if graph:
                    r16 = _GraphStep(diff, model, x16.clone(), kw16, True, None)
                    step16 = lambda k: r16.step(k % T)
                else:
                    st16 = {"img": x16.clone()}
                    tab16 = diff._step_table(dev, n16)

                    def step16(k):
                        st16["img"] = diff.ddim_sample(model, st16["img"], tab16[k % T], model_kwargs=kw16)["sample"]
                for k in range(3):
                    step16(k)
                ts = []
                for r in range(3):
                    sync()
                    t0 = time.perf_counter()
                    for k in range(k16):
                        step16(3 + r * k16 + k)
                    sync()
                    ts.append(time.perf_counter() - t0)
                return statistics.median(ts)
            d16, e16 = run16(True), run16(False)
            extra["batch16"] = {"value": n16 * k16 / d16, "unit": "image-steps/s", "ms_per_step": 1e3 * d16 / k16, "steps": k16, "regions": 3,
                                "eager_ms_per_step": 1e3 * e16 / k16, "eager_image_steps_per_sec": n16 * k16 / e16,
                                "model_tflops": n16 * k16 / d16 * GFLOP_PER_IMAGE_STEP_P64 / 1e3,
                                "frac_of_roof": n16 * k16 / d16 * GFLOP_PER_IMAGE_STEP_P64 / 1e3 / (F16_MFMA_PEAK_TFLOPS / 3.0 if prec0 == "f16x3" else FP32_MFMA_PEAK_TFLOPS),
                                "workload": "the same P64 DDIM step at batch 16 (SURVEY 8d config 3, N = 16): graph replay; eager launches beside it"}
        # secondary leg (single GPU): the other product mode — IEEE fp32 products on v_mfma_f32_32x32x2_f32 against its own roof
        other = None
        if single and not args.no_fp32:
            alt = "fp32" if prec0 == "f16x3" else "f16x3"
            causaldiffae_amd.set_precision(alt)
            try:
                k2, r2 = max(20, min(args.steps, 30)), 3           # the headline's rigor: >= 20 steps x 3 regions, median region
                s2 = timed_ddim(k2, 3, r2)
                dt2 = statistics.median(s2)
                other = {"precision_mode": alt, "value": N * k2 / dt2, "unit": "image-steps/s", "ms_per_step": 1e3 * dt2 / k2, "steps": k2,
                         "regions": r2, "warmup": 3, "spread": spread([N * k2 / s_ for s_ in s2]),
                         "model_tflops": N * k2 / dt2 * GFLOP_PER_IMAGE_STEP_P64 / 1e3, "roofline": roofline(alt, 2)}
            finally:
                causaldiffae_amd.set_precision(prec0)

    if rank != 0:
        return
    value = world * N * args.steps / dt
    rates = [world * N * args.steps / s for s in region_s]
    dtype_of = {"fp32": "f32 (IEEE fp32 products, v_mfma_f32_32x32x2_f32)",
                "f16x3": "f32 in/out/accumulate; products as f16x3 split (hi*hi + hi*lo + lo*hi on f16 MFMA, 2^-22 relative)"}
    out = {
        "metric": "DDIM denoise image-steps/sec, 64x64 UNet (P64)", "value": value, "unit": "image-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype_of[prec0], "data": "synthetic",
        "config": {"workload": "Pendulum 64x64 C=4, 4 causal vars, DDIM-100 counterfactual sampling (encode -> intervene -> "
                               "q_sample -> ddim steps), UNet 93.45M params", "batch_per_gpu": N, "global_batch": N * world,
                   "parallelism": f"batch-sharded x{world}, no collectives", "hip_graph": not args.no_graph},
        "timed_regions": {"each": f"exactly {args.steps} steps", "value_is": "median region", **spread(rates)},
        "dist_backend": (dist.get_backend() if world > 1 else None), "ranks_in_group": (dist.get_world_size() if world > 1 else 1),
        "samples_per_sec_ddim100": value / 100.0,
        "model_tflops": value * GFLOP_PER_IMAGE_STEP_P64 / 1e3,
        "roofline": roof,
        "box": box,
        # the whole step against the nominal roof and against this box's sustained MFMA rate (f16x3: / 3)
        "whole_step_frac_of_roof": value * GFLOP_PER_IMAGE_STEP_P64 / 1e3 / world / (F16_MFMA_PEAK_TFLOPS / 3.0 if prec0 == "f16x3" else FP32_MFMA_PEAK_TFLOPS),
        "whole_step_frac_of_sustained_mfma": (value * GFLOP_PER_IMAGE_STEP_P64 / 1e3 / world / (box["mfma_sustained_tflops"] / 3.0)) if box and prec0 == "f16x3" else None,
        "other_precision": other,
        "train": train,
        **extra,
    }
    if not args.no_cpu_baseline and single:
        out["cpu_baseline"] = cpu_baseline()
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if train and "value" in train:
            out["train_gpu_over_cpu_images_per_sec"] = train["images_per_sec"] / out["cpu_baseline"]["train"]["images_per_sec"]
    # The full record (per-family milliseconds, per-thread CPU lists, spreads of every leg) goes to a file; the printed line carries the
    # contract keys + roofline + cpu_baseline + the headline number of every leg, the training block LAST, and stays well under 8 KB.
    full_path = os.environ.get("CDAE_BENCH_FULL", os.path.join(ROOT, "gpurun_out", "bench_full.json"))
    try:
        os.makedirs(os.path.dirname(full_path), exist_ok=True)
        with open(full_path, "w") as f:
            json.dump(out, f)
    except OSError:
        full_path = None
    line = compact(out, full_path)
    print(json.dumps(line))


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _r(x, nd=4):
    """floats to `nd` significant digits (the file keeps full precision), containers recursively"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def compact(out, full_path):
    """The one JSON line of the contract: every required key, `roofline`, `cpu_baseline`, then one short block per secondary leg with its
    headline numbers — `train` (the second half of BASELINE's metric) printed last so that a tail of the output always holds it."""
    head = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                       "dtype", "data", "config", "samples_per_sec_ddim100", "model_tflops", "whole_step_frac_of_roof",
                       "whole_step_frac_of_sustained_mfma", "dist_backend", "ranks_in_group"))
    head["timed_regions"] = _pick(out.get("timed_regions") or {}, ("each", "value_is", "median", "min", "max", "regions"))
    rf = out.get("roofline")
    if rf:
        r = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_unit", "traffic_source", "traffic_measured_in_run",
                       "algorithmic_bytes_per_launch", "traffic_ratio", "precision_mode", "launches_per_step", "avg_launch_us",
                       "flops_per_launch_avg", "share_of_contraction_time", "frac_of_sustained_mfma", "flops_convention"))
        r["kernel"] = r.get("kernel", "")[:120]
        if "all_contractions" in rf:
            r["all_contractions"] = _pick(rf["all_contractions"], ("achieved", "frac", "launches_per_step", "ms_per_step"))
        if "upconv_4tap" in rf:
            r["upconv_4tap"] = _pick(rf["upconv_4tap"], ("kernel", "frac", "avg_launch_us", "ms_per_step", "traffic", "algorithmic_bytes_per_launch", "traffic_ratio"))
        if "hbm_kernels" in rf:
            r["hbm_kernels"] = rf["hbm_kernels"]
        head["roofline"] = r
    head["box"] = out.get("box")
    cb = out.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "sample"))
        c["train"] = _pick(cb.get("train", {}), ("value", "unit", "images_per_sec"))
        c["config0_m32"] = _pick(cb.get("config0_m32", {}), ("p_sample_image_steps_per_sec", "train_steps_per_sec"))
        head["cpu_baseline"] = c
        head.update(_pick(out, ("gpu_over_cpu", "train_gpu_over_cpu_images_per_sec")))
    if out.get("public_ddim_sample_loop"):
        head["public_ddim_sample_loop"] = _pick(out["public_ddim_sample_loop"], ("steps_per_call", "default_ms_per_step", "eager_ms_per_step", "default_image_steps_per_sec"))
    for k in ("guided_w2", "batch16"):
        if out.get(k):
            head[k] = _pick(out[k], ("value", "unit", "ms_per_step", "eager_ms_per_step", "frac_of_roof", "forwards_per_step"))
    op = out.get("other_precision")
    if op:
        o = _pick(op, ("precision_mode", "value", "unit", "ms_per_step"))
        if op.get("roofline"):
            o["roofline"] = _pick(op["roofline"], ("kernel", "achieved", "peak", "frac", "traffic", "traffic_ratio", "avg_launch_us"))
            o["roofline"]["kernel"] = o["roofline"].get("kernel", "")[:60]
        head["other_precision"] = o
    head["full_record"] = (os.path.relpath(full_path, ROOT) if full_path else None)
    tr = out.get("train")
    if tr:
        leg = ("value", "unit", "ms_per_step", "batch_per_gpu", "global_batch", "images_per_sec", "model_tflops", "roof_tflops", "frac_of_roof", "precision_mode",
               "last_loss", "host_cpu_ms_per_step", "host_cpu_over_step", "host_cpu_quota_cores", "host_bound_risk", "host_cores_needed_at_world8",
               "host_bound_risk_at_world8", "wgrad_side_stream", "launch_mode", "error")
        t = _pick(tr, leg + ("workload", "steps", "warmup", "dist_backend"))
        t["spread"] = _pick(tr.get("spread", {}), ("min", "max", "regions"))
        t["host_cpu_ms_per_step_by_thread"] = (tr.get("host_cpu_ms_per_step_by_thread") or [])[:3]
        w8keys = ("value", "ms_per_step", "host_cpu_ms_per_step", "host_cpu_over_step", "wgrad_side_stream")
        if tr.get("world8_policy"):
            t["world8_policy"] = _pick(tr["world8_policy"], w8keys)
        if "fp32_mode" in tr:
            t["fp32_mode"] = _pick(tr["fp32_mode"], ("value", "ms_per_step", "frac_of_roof", "precision_mode"))
        if "config1_m32_b256" in tr:
            c1 = tr["config1_m32_b256"]
            t["config1_m32_b256"] = {k: {**_pick(c1[k], leg), "world8_policy": _pick(c1[k].get("world8_policy") or {}, w8keys)} for k in ("f16x3", "mixed16") if k in c1}
            t["config1_m32_b256"].update(_pick(c1, ("mixed16_over_f16x3",)))
        if "mixed16_torso" in tr:
            t["mixed16_torso"] = {**_pick(tr["mixed16_torso"], leg), "world8_policy": _pick(tr["mixed16_torso"].get("world8_policy") or {}, w8keys)}
            t.update(_pick(tr, ("mixed16_torso_over_f16x3",)))
        head["train"] = t
    return _r(head, 5)


if __name__ == "__main__":
    main()
