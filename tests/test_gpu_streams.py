"""Stream order links (include/cdae.h cdae_stream_link_*): the cross-stream ordering the trainer's weight-gradient side stream uses
instead of hipEvent record / wait (reference train_util.py:255-259 runs backward on one stream; the overlap is this implementation's).
Checked here: the ordering itself (a dependent producer / consumer chain over two streams, both directions, thousands of hand-offs),
that the training step is bit-identical with links and with events, and that a link costs the host no spinning helper thread."""
import ctypes
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _links():
    from causaldiffae_amd._lib import check, lib
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    check(lib.cdae_stream_link_create(ctypes.byref(a)))
    check(lib.cdae_stream_link_create(ctypes.byref(b)))
    return a, b


def test_stream_link_orders_a_dependent_chain():
    """main: x += 1 (a 64 MB tensor: tens of microseconds) -> link -> side: acc += x (reads what main just wrote) -> link back -> main: x += 1 ...
    With the ordering enforced acc = 1 + 2 + ... + n in EVERY element; a consumer that ran early or a producer that overwrote x before
    the consumer had read it gives another sum."""
    from causaldiffae_amd._lib import check, lib
    fork, join = _links()
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    x = torch.zeros(1 << 24, device=DEV)
    acc = torch.zeros(1 << 24, device=DEV, dtype=torch.float64)
    n = 600
    torch.cuda.synchronize()
    for i in range(n):
        x.add_(1.0)
        check(lib.cdae_stream_link_order(fork, main.cuda_stream, side.cuda_stream))
        with torch.cuda.stream(side):
            acc.add_(x)
        check(lib.cdae_stream_link_order(join, side.cuda_stream, main.cuda_stream))
    torch.cuda.synchronize()
    want = n * (n + 1) / 2
    assert float(acc.min()) == want and float(acc.max()) == want, (float(acc.min()), float(acc.max()), want)
    # one direction only, many producers in flight: the consumer may lag, never lead
    y = torch.zeros(1 << 22, device=DEV)
    seen = torch.zeros(256, device=DEV)
    for i in range(256):
        y.fill_(float(i + 1))
        check(lib.cdae_stream_link_order(fork, main.cuda_stream, side.cuda_stream))
        with torch.cuda.stream(side):
            seen[i:i + 1].copy_(y[-1:])
        check(lib.cdae_stream_link_order(join, side.cuda_stream, main.cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(seen.cpu(), torch.arange(1, 257, dtype=torch.float32))
    check(lib.cdae_stream_link_destroy(fork))
    check(lib.cdae_stream_link_destroy(join))


def _thread_cpu():
    tick, out = os.sysconf("SC_CLK_TCK"), {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (int(rest[11]) + int(rest[12])) / tick
        except (OSError, ValueError):
            pass
    return out


def test_stream_link_keeps_no_host_thread_spinning():
    """The reason the links exist: with hipEvent record / wait per hand-off a runtime helper thread is busy for as long as a dependency is
    pending (~1 core); with links every thread but the launching one stays (nearly) idle.  Measured as CPU seconds per second of wall time
    of all threads other than the caller's."""
    from causaldiffae_amd._lib import check, lib
    fork, join = _links()
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    x = torch.zeros(1 << 22, device=DEV)
    y = torch.zeros(1 << 22, device=DEV)
    me = int(open("/proc/thread-self/stat").read().split()[0]) if os.path.exists("/proc/thread-self/stat") else os.getpid()

    def others_busy(use_links, n=6000):
        torch.cuda.synchronize()
        t0, w0 = _thread_cpu(), time.perf_counter()
        for i in range(n):
            x.add_(1.0)
            if use_links:
                check(lib.cdae_stream_link_order(fork, main.cuda_stream, side.cuda_stream))
            else:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                y.add_(1.0)
            if use_links:
                check(lib.cdae_stream_link_order(join, side.cuda_stream, main.cuda_stream))
            else:
                main.wait_stream(side)
            if i % 256 == 255:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        w, t1 = time.perf_counter() - w0, _thread_cpu()
        return sum(v - t0.get(k, 0.0) for k, v in t1.items() if k != me) / w

    others_busy(True, 500)
    with_links, with_events = others_busy(True), others_busy(False)
    print(f"other threads busy: links {with_links:.2f} cores, events {with_events:.2f} cores")
    assert with_links < 0.35, (with_links, with_events)
    check(lib.cdae_stream_link_destroy(fork))
    check(lib.cdae_stream_link_destroy(join))


def test_training_step_identical_with_links_and_with_events():
    """One C64 training step (batch 4, weight gradients on the side stream) ordered by stream links and by events: the same gradients
    bit for bit — the links change how the streams are ordered, not what runs."""
    import bench
    from causaldiffae_amd import ops
    from improved_diffusion import script_util as su
    from improved_diffusion.train_util import TrainLoop
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(11)
    N = 4
    x0 = torch.rand(N, 3, 64, 64, generator=g) * 2 - 1
    cond = {"c": torch.rand(N, 4, generator=g)}
    t = torch.randint(0, 1000, (N,), generator=g)
    noise = torch.randn(N, 3, 64, 64, generator=g)

    def run(links):
        cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True}
        model, diff = su.create_model_and_diffusion(**cfg)
        bench.randomize(model, 4321)
        model.to(dev).train()
        loop = TrainLoop(model=model, diffusion=diff, data=iter(()), batch_size=N, microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9,
                         save_interval=10 ** 9, resume_checkpoint="", rep_cond=True, n_vars=4, causal_modeling=True, in_channels=3)
        diff.kl_weight = 0.1
        with ops.path_scope(stream_links=links, wgrad_stream=True):
            for _ in range(2):
                loop.opt.zero_grad()
                torch.manual_seed(9)
                losses = diff.training_losses(model, x0.to(dev), t.to(dev), model_kwargs={k: v.to(dev) for k, v in cond.items()}, noise=noise.to(dev),
                                              rep_cond=True, causal_modeling=True)
                losses["loss"].mean().backward()
                ops.side_join()
        torch.cuda.synchronize()
        return losses["loss"].detach().clone(), loop.opt.flat.grad.clone()

    la, ga = run(True)
    lb, gb = run(False)
    assert torch.equal(la, lb)
    assert torch.equal(ga, gb), float((ga - gb).abs().max())
    assert float(ga.abs().max()) > 0
