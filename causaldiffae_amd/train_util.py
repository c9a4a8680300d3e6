"""Training runtime: TrainLoop with the reference's constructor / semantics (improved_diffusion/train_util.py:
microbatching, KL-weight warm-up, lr anneal, EMA, checkpoint names) on an MI355X-native data path:

  * all parameters live in ONE flat fp32 buffer (the nn.Parameters are views), gradients in a second one:
    AdamW + EMA is a single fused kernel over the flat buffer (reference: ~384 tensors x several ATen ops);
  * data parallel = one process per GPU; gradients are all-reduced in contiguous buckets of the flat
    gradient buffer over RCCL/xGMI, each bucket launched asynchronously from a post-accumulate hook as soon as
    its last gradient is ready, i.e. overlapped with the rest of backward (DDP-style, train_util.py:111-118);
  * rank 0 broadcasts parameters, EMA and BN buffers at start (the reference's sync_params is a no-op, Q6).
"""
import copy
import os
import time

import numpy as np
import torch as th
import torch.distributed as dist

from . import dist_util, logger, ops
from ._lib import check, lib, ptr, stream
from .resample import LossAwareSampler, UniformSampler


FLAT_ALIGN = 8         # elements: see FlatParams.__init__


class FlatParams:
    """Re-homes a module's parameters (and their .grad) as views of two flat fp32 buffers, in registration order."""

    def __init__(self, model):
        named = list(model.named_parameters())
        # the ResBlocks' emb_layers weights (then their biases) first and adjacent: their concatenation is then a VIEW of the flat
        # buffer and the batched embedding GEMM (ops._EmbAllTrain) needs no copies, forward or backward.  Being first they are also
        # the last bucket to be all-reduced, which matches when their gradient is produced (the very end of backward).
        group = model.emb_param_groups() if hasattr(model, "emb_param_groups") else None
        lead = []
        if group is not None and group[1] and len({w.shape[1] for w in group[1]}) == 1 and all(w.is_contiguous() for w in group[1]):
            lead = list(group[1]) + list(group[2])
            ids = {id(p) for p in lead}
            by_id = {id(p): n for n, p in named}
            named = [(by_id[id(p)], p) for p in lead] + [(n, p) for n, p in named if id(p) not in ids]
        self.params = [p for _, p in named]
        self.names = [n for n, _ in named]
        dev = self.params[0].device
        # every view starts on a 32-byte boundary (8 fp32 = 8 bf16 of the buffer's 16-bit image = one 16-byte piece of it): the streaming
        # kernels read weights / write weight gradients as 16-byte pieces, and n_vars = 3 or 5 makes `causal_mask.*.bias` 170 / 102 elements,
        # which would leave every 1x1 weight behind it 8 bytes off.  The padding elements are zero in weights, gradients and moments and stay zero.
        self.offsets, off = [], 0
        for p in self.params:
            off = (off + FLAT_ALIGN - 1) // FLAT_ALIGN * FLAT_ALIGN
            self.offsets.append(off)
            off += p.numel()
        self.numel = off
        self.flat = th.zeros(off, dtype=th.float32, device=dev)
        self.grad = th.zeros(off, dtype=th.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            view = self.flat.as_strided(p.shape, p.stride(), o)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad.as_strided(p.shape, p.stride(), o)
            # backward kernels accumulate straight into this view (ops._sink): no temporaries, no autograd add kernels
            p._grad_view = p.grad
            p._grad_ready = None
        emb_w = None
        if lead:
            blocks, ws, bs = group
            T, K = sum(w.shape[0] for w in ws), ws[0].shape[1]
            offs, o = [], 0
            for blk, w in zip(blocks, ws):
                offs.append((id(blk), o, w.shape[0]))
                o += w.shape[0]
            assert self.offsets[len(ws)] == T * K and all(self.offsets[len(ws) + i + 1] - self.offsets[len(ws) + i] == w.shape[0] for i, w in enumerate(ws[:-1])), \
                "emb_layers weights / biases must lie back to back (channel counts that are multiples of 8)"
            emb_w = self.flat[:T * K].view(T, K)
            model._emb_flat = dict(w=emb_w, b=self.flat[T * K:T * K + T], gw=self.grad[:T * K].view(T, K),
                                   gb=self.grad[T * K:T * K + T], params=lead, offs=offs)
        # power-of-two scale records of every weight the f16 modes hand to the matrix cores (ops.weight_scale): one pass per weight
        # version over the flat buffer, the concatenated emb_layers weight (ONE GEMM operand) as a tensor of its own; then the operand
        # planes of the 3x3 convs (which consume their records)
        self.scale_table = ops.register_scale_table(self.flat, self.params + ([emb_w] if emb_w is not None else [])) if dev.type == "cuda" else None
        self.conv_bank = ops.register_conv_bank(self.flat, self.params) if dev.type == "cuda" else None
        if dev.type == "cuda":
            ops.register_flat16(self.flat, self.params)          # the 16-bit torso's 1x1 / linear weights: one bf16 image of the buffer per weight version

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):          # re-attach if someone set .grad = None
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad.as_strided(p.shape, p.stride(), o)
                p._grad_view = p.grad


class GradBuckets:
    """Contiguous buckets over the flat gradient buffer, reduced asynchronously as they become ready.

    Buckets are built from the END of the parameter list (gradients arrive roughly in reverse registration
    order during backward).  xGMI is point-to-point: ring all-reduce time is per-link bound, so a few large
    buckets (default 64 MiB) amortise latency while still overlapping with the remaining backward."""

    def __init__(self, flat, bucket_bytes=64 << 20, group=None):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # the bucketed all-reduce runs for world > 1 — and, CDAE_DDP_FORCE=1, in an initialised process group of ONE rank: the whole
        # data-parallel path (hooks, ordered bucket launches, RCCL's stream, the waits) on a one-GPU box with the real backend, where the
        # reduction is the identity (tests/test_gpu_model.py::test_rccl_data_parallel_path_with_one_rank)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("CDAE_DDP_FORCE") == "1")
        self.launched = 0           # collectives issued (diagnostics)
        self._gloo = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self.buckets, self.bucket_of = [], {}
        hi, acc, members = flat.numel, 0, []
        for i in range(len(flat.params) - 1, -1, -1):
            members.append(i)
            acc += flat.params[i].numel() * 4
            if acc >= bucket_bytes or i == 0:
                lo = flat.offsets[i]
                self.buckets.append(dict(lo=lo, hi=hi, members=list(members), pending=0, work=None))
                for m in members:
                    self.bucket_of[m] = len(self.buckets) - 1
                hi, acc, members = lo, 0, []
        self.enabled = True
        self._hooks = []
        if self.active:
            for i, p in enumerate(flat.params):
                if p.requires_grad:
                    hook = self._make_hook(i)
                    self._hooks.append(p.register_post_accumulate_grad_hook(hook))    # gradients that arrive through autograd
                    p._grad_ready = (lambda h=hook, q=p: h(q))                         # gradients written in place by ops._sink
        self.reset()

    def reset(self):
        for b in self.buckets:
            b["pending"] = sum(1 for m in b["members"] if self.flat.params[m].requires_grad)
            b["work"] = None
        self.fired = [False] * len(self.flat.params)
        self.next_launch = 0          # collectives are issued in bucket-index order on EVERY rank, whatever order the hooks fire in

    def _launch_ready(self):
        """Launch every complete bucket whose predecessors have been launched.  A rank whose gradients arrive in a different order
        (e.g. a parameter that falls back from the in-place sink to the autograd path) must still issue the same sequence of
        collectives as its peers, or RCCL pairs up different buckets and deadlocks — the rule DDP follows."""
        while self.next_launch < len(self.buckets) and self.buckets[self.next_launch]["pending"] == 0:
            b = self.buckets[self.next_launch]
            if b["work"] is None:
                b["work"] = self._all_reduce(b)
            self.next_launch += 1

    def _all_reduce(self, b):
        """Asynchronous all-reduce of one bucket.  Ordering behind the kernels that wrote it is same-stream ordering: the hooks
        (`_grad_ready` from ops._done and autograd's post-accumulate hook) run on the thread that enqueued the backward kernels —
        autograd's device thread, whose current stream is the forward's stream — and the C-ABI launches go to exactly that stream
        (`_lib.stream()` = torch's current stream of the calling thread); the process group makes its own stream wait on the current
        stream before the collective starts.  No event of our own: one recorded here would be recorded on, and waited for by, the
        same stream."""
        grad = self.flat.grad[b["lo"]:b["hi"]]
        ops.side_join()             # weight gradients are written on ops' side stream: this stream waits for them before the collective reads them
        if self._gloo and grad.is_cuda:
            # gloo (the stand-in for RCCL on a one-GPU test box) stages the bucket through pinned memory on a pool stream that waits for an
            # event of the launch stream.  With four processes time-slicing ONE GPU and no other cross-stream traffic (side stream off) that
            # wait never returned from the second step on (every rank's launch stream stalled, found with tools/hang_bt.sh; two ranks, or
            # AMD_SERIALIZE_KERNEL=3, or the side stream's own waits make it go away).  RCCL runs its kernels on the device and has no such
            # staging copy; for gloo the host simply waits for the launch stream first.
            th.cuda.current_stream(grad.device).synchronize()
        self.launched += 1
        return dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _make_hook(self, i):
        def hook(_p):
            # a parameter whose gradient is written in place (ops._sink) reports through `_grad_ready`, and autograd's
            # post-accumulate hook fires for it AS WELL although the Function returned None for it: count each parameter once,
            # or a bucket is all-reduced while half of its gradients are still missing
            if not self.enabled or self.fired[i]:
                return
            self.fired[i] = True
            b = self.buckets[self.bucket_of[i]]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch_ready()
        return hook

    def reduce_all(self):
        """All buckets at once, after a backward that ran with the hooks disabled (hipGraph replay)."""
        if not self.active:
            return
        self.reset()
        self.finish()

    def finish(self, average=True):
        """Wait for in-flight buckets, reduce any bucket whose hook never fired (unused params), average.  `average=False` leaves the
        all-reduced SUM in the flat buffer: TrainLoop hands 1 / world to the optimizer kernel's gradient scale instead of spending a
        pass of its own over the buffer (374 MB for the 93 M-parameter UNet)."""
        if not self.active:
            return
        for b in self.buckets:             # in index order, like the hooks
            if b["work"] is None:
                b["work"] = self._all_reduce(b)
        for i, b in enumerate(self.buckets):
            try:
                b["work"].wait()
            except Exception as e:
                import sys
                print(f"[rank {dist.get_rank()}] bucket {i} of {len(self.buckets)} [{b['lo']}:{b['hi']}] wait failed: {e!r}; completed: "
                      f"{[bb['work'].is_completed() for bb in self.buckets]}", file=sys.stderr, flush=True)
                raise
        if average:
            self.flat.grad.mul_(1.0 / self.world)
        self.reset()


class FusedAdamWEMA:
    """torch.optim.AdamW semantics + update_ema (reference train_util.py:292-297, nn.py:503-513) as ONE kernel over
    the flat parameter buffer per EMA rate."""

    def __init__(self, model, lr=1e-4, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, ema_rates=(0.9999,)):
        self.model = model
        self.flat = FlatParams(model)
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.ema_rates = list(ema_rates)
        self.m = th.zeros_like(self.flat.flat)
        self.v = th.zeros_like(self.flat.flat)
        self.ema = [self.flat.flat.clone() for _ in self.ema_rates]
        self.t = 0
        self._sq = th.zeros(1, dtype=th.float64, device=self.flat.flat.device)

    def zero_grad(self):
        ops.side_join()                  # (no weight-gradient launch of the previous step may still be accumulating: see ops.side_launch)
        self.flat.zero_grad()

    def grad_sqsum(self, grad_scale=1.0):
        """squared norm of (grad_scale * gradient)"""
        ops.side_join()
        check(lib.cdae_sqsum(ptr(self.flat.grad), self.flat.numel, ptr(self._sq), stream()))
        return float(self._sq.item()) * grad_scale * grad_scale

    def step(self, lr=None, grad_scale=1.0):
        """One AdamW step on `grad_scale * gradient` + every EMA rate, ONE kernel over the flat buffers (groups of four rates per launch;
        only the first launch of a step moves the weights)."""
        import ctypes
        self.t += 1
        f = self.flat
        ops.side_join()                  # backward() already joined ops' weight-gradient stream; a caller that wrote gradients otherwise has not
        n = len(self.ema)
        emas = (ctypes.c_void_p * max(1, min(4, n)))(*[e.data_ptr() for e in self.ema[:4]])
        rates = (ctypes.c_double * max(1, min(4, n)))(*self.ema_rates[:4])
        check(lib.cdae_adamw_ema_multi(ptr(f.flat), ptr(f.grad), ptr(self.m), ptr(self.v), emas, rates, min(4, n), f.numel,
                                       self.lr if lr is None else lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                       self.t, grad_scale, stream()))
        for rate, e in zip(self.ema_rates[4:], self.ema[4:]):          # (more than four rates: never seen in the reference's scripts)
            e.mul_(rate).add_(f.flat, alpha=1 - rate)
        ops.bump_weight_epoch()          # parameter storage was rewritten by a raw kernel: cached pre-split weight planes are stale

    def ema_state_dict(self, i):
        """EMA weights under the model's parameter names / shapes (buffers copied from the live model)."""
        sd = self.model.state_dict()
        f = self.flat
        for n, p, o in zip(f.names, f.params, f.offsets):
            sd[n] = self.ema[i].as_strided(p.shape, p.stride(), o)
        return sd

    def torch_state_dict(self):
        """The Adam moments in torch.optim.AdamW.state_dict() layout over model.parameters() order — the format the reference's trainer
        writes and loads under `opt<step>.pt` (train_util.py:159-169, 340-343), so either trainer resumes from the other's file."""
        f = self.flat
        where = {n: (p, o) for n, p, o in zip(f.names, f.params, f.offsets)}
        order = [n for n, _ in self.model.named_parameters()]
        view = lambda buf, n: buf.as_strided(where[n][0].shape, where[n][0].stride(), where[n][1]).detach().cpu().contiguous()
        state = {i: {"step": th.tensor(float(self.t)), "exp_avg": view(self.m, n), "exp_avg_sq": view(self.v, n)} for i, n in enumerate(order)} if self.t > 0 else {}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def broadcast_from_rank0(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.flat.flat, 0)
            for e in self.ema:
                dist.broadcast(e, 0)
            for b in self.model.buffers():
                dist.broadcast(b, 0)
        ops.bump_weight_epoch()


def linear_kl_weight(step, total_steps=50000, initial=0.0, final=1.0):
    """KL warm-up (reference train_util.py:176-187): linear over total_steps with t = step/(total_steps-1)."""
    if step >= total_steps:
        return final
    if step <= 0:
        return initial
    if total_steps <= 1:
        return final
    t = step / (total_steps - 1)
    return (1.0 - t) * initial + t * final


class TrainLoop:
    def __init__(self, *, model, diffusion, data, batch_size, microbatch, lr, ema_rate, log_interval, save_interval,
                 resume_checkpoint, use_fp16=False, fp16_scale_growth=1e-3, schedule_sampler=None, weight_decay=0.0,
                 lr_anneal_steps=0, rep_cond=False, n_vars=None, causal_modeling=False, flow_based=False, in_channels=3,
                 masking=False, bucket_mb=64, use_graph=False, run_ahead=1, throttle_poll_s=0.0005):
        if use_fp16:       # reduced-precision torso; fp32 master weights are the only weights, bf16 gradients need no loss scaling
            model.convert_to_fp16()        # marks THIS model (the mode is scoped to its forward / backward, not process-wide)
        self.model, self.diffusion, self.data = model, diffusion, data
        self.batch_size = batch_size
        self.microbatch = microbatch if microbatch > 0 else batch_size
        self.lr = lr
        self.ema_rate = [ema_rate] if isinstance(ema_rate, float) else [float(x) for x in ema_rate.split(",")]
        self.log_interval, self.save_interval, self.resume_checkpoint = log_interval, save_interval, resume_checkpoint
        self.schedule_sampler = schedule_sampler or UniformSampler(diffusion)
        self.weight_decay, self.lr_anneal_steps = weight_decay, lr_anneal_steps
        self.rep_cond, self.n_vars, self.causal_modeling = rep_cond, n_vars, causal_modeling
        self.flow_based, self.in_channels, self.masking = flow_based, in_channels, masking
        self.step, self.resume_step = 0, 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.global_batch = self.batch_size * self.world

        if resume_checkpoint:
            self.resume_step = parse_resume_step_from_filename(resume_checkpoint)
            self.model.load_state_dict(dist_util.load_state_dict(resume_checkpoint, map_location="cpu"))
        self.opt = FusedAdamWEMA(model, lr=lr, weight_decay=weight_decay, ema_rates=self.ema_rate)
        if resume_checkpoint:
            self._load_resume_state(resume_checkpoint)
        self.opt.broadcast_from_rank0()
        self.buckets = GradBuckets(self.opt.flat, bucket_bytes=bucket_mb << 20)
        self.model_params = self.opt.flat.params
        self.master_params = self.model_params
        self.use_ddp = self.world > 1
        self.ddp_model = self.model
        self.last_losses = None
        self.use_graph = bool(use_graph)
        self._graphs, self._graph_failed, self._eager_steps = {}, False, 0
        # run_ahead: optimizer steps the host may enqueue beyond the one the GPU executes (None: unbounded); see _throttle
        self.run_ahead, self._step_events = (None if run_ahead is None else max(1, int(run_ahead))), []
        self.throttle_poll_s = float(throttle_poll_s)

    # ------------------------------------------------------------------ loop
    def run_loop(self):
        while not self.lr_anneal_steps or self.step + self.resume_step < self.lr_anneal_steps:
            batch, cond = next(self.data)
            self.run_step(batch, cond)
            if self.step % self.log_interval == 0:
                logger.dumpkvs()
            if self.step % self.save_interval == 0:
                self.save()
                if os.environ.get("DIFFUSION_TRAINING_TEST", "") and self.step > 0:
                    return
            self.step += 1
            self.diffusion.kl_weight = linear_kl_weight(self.step, 50000, 0.0, 1.0)     # applied AFTER the step (train_util.py:210-214)
        if (self.step - 1) % self.save_interval != 0:
            self.save()

    def run_step(self, batch, cond):
        self.forward_backward(batch, cond)
        self.optimize_normal()
        self.log_step()

    # ------------------------------------------------------------------ hipGraph replay of forward + backward
    def _graph_wanted(self, batch):
        """A training step is ~1600 kernel launches, many of them a few microseconds long: launched one by one the host, not the
        GPU, sets the step time.  With static shapes (one microbatch, uniform timestep sampling) the whole forward + backward is
        captured once into a hipGraph and replayed; inputs, timesteps, loss weights and the KL weight live in static buffers.
        The gradient all-reduce (world > 1) and the optimizer kernel stay outside the graph.
        OPT-IN (use_graph=True): gradients match the eager step to its own run-to-run noise (6e-6), but replay is no faster on ROCm 7.2.
        Re-measured in round 6 on today's step (C64 batch 32, ~1080 nodes, same MI355X box, tools/exp_train.py GRAPH=1): parity mode
        28.1 ms replayed against 26.9 ms eager, 16-bit torso 18.1 against 16.8 (round 1, 1600 nodes: 48.9 against 42.0).  The eager launch
        queue already runs ahead of the GPU and overlaps the weight-gradient stream with the data-gradient chain; the captured graph keeps
        the fork as a branch but its nodes do not overlap.  Nor is replay free for the host: hipGraphLaunch walks the nodes on the CPU
        (33 ms of process CPU per replayed step against 46 eager)."""
        return (self.use_graph and not self._graph_failed and th.cuda.is_available() and isinstance(self.schedule_sampler, UniformSampler)
                and self.microbatch >= batch.shape[0] and self._eager_steps >= 2)

    def _graph_capture(self, batch, cond):
        dev = dist_util.dev()
        st = dict(batch=batch.to(dev).clone(), cond={k: v.to(dev).clone() for k, v in cond.items()},
                  t=th.zeros(batch.shape[0], dtype=th.int64, device=dev), w=th.ones(batch.shape[0], dtype=th.float32, device=dev),
                  klw=th.zeros((), dtype=th.float32, device=dev))
        kl_saved = self.diffusion.kl_weight
        self.diffusion.kl_weight = st["klw"]
        self.buckets.enabled = False
        ops.bump_weight_epoch()              # the captured step must contain the weight-plane kernels, not reuse cached planes
        graph = th.cuda.CUDAGraph()
        try:
            with th.cuda.graph(graph):
                self.opt.zero_grad()
                losses = self.diffusion.training_losses(self.model, st["batch"], st["t"], model_kwargs=dict(st["cond"]),
                                                        rep_cond=self.rep_cond, causal_modeling=self.causal_modeling)
                loss = (losses["loss"] * st["w"]).mean()
                loss.backward()
            st["losses"] = {k: v.detach() for k, v in losses.items()}
            st["graph"] = graph
        finally:
            self.diffusion.kl_weight = kl_saved
            ops.bump_weight_epoch()          # planes cached during capture belong to the graph's private pool
        return st

    def _graph_step(self, batch, cond):
        key = (tuple(batch.shape), tuple(sorted((k, tuple(v.shape)) for k, v in cond.items())))
        st = self._graphs.get(key)
        if st is None:
            try:
                st = self._graphs[key] = self._graph_capture(batch, cond)
            except Exception as e:          # capture is an optimisation: anything it cannot record runs eagerly, loudly
                self._graph_failed = True
                logger.log(f"hipGraph capture of the training step failed ({type(e).__name__}: {e}); running eagerly")
                return False
        dev = dist_util.dev()
        st["batch"].copy_(batch, non_blocking=True)
        for k, v in cond.items():
            st["cond"][k].copy_(v, non_blocking=True)
        t, weights = self.schedule_sampler.sample(batch.shape[0], dev)
        st["t"].copy_(t)
        st["w"].copy_(weights)
        st["klw"].fill_(float(self.diffusion.kl_weight))
        st["graph"].replay()
        self.last_losses, self.last_t, self.last_w = st["losses"], st["t"], st["w"]
        self.buckets.reduce_all()
        return True

    @property
    def grad_scale(self):
        """What to multiply `p.grad` / the flat gradient buffer by to get the data-parallel MEAN gradient (see forward_backward)."""
        return getattr(self, "_grad_scale", 1.0)

    def forward_backward(self, batch, cond):
        """Contract (differs from the reference's DDP, which averages before the optimizer): after this call the flat gradient buffer — and so
        every `p.grad` — holds the cross-rank SUM; the 1 / world factor is `self.grad_scale`, which `optimize_normal` hands to the fused
        AdamW / EMA kernel and `log_step` to the gradient norm.  Anything else that reads `p.grad` between the two calls (clipping, custom
        logging, another optimizer) must multiply by `self.grad_scale`."""
        from ._lib import precision_scope
        with precision_scope(getattr(self.model, "_cdae_precision", None)):      # the backward kernels run in the model's own mode too
            self._forward_backward(batch, cond)

    def _forward_backward(self, batch, cond):
        if self._graph_wanted(batch) and self._graph_step(batch, cond):
            self._grad_scale = 1.0           # (reduce_all averaged)
            return
        self._eager_steps += 1
        dev = dist_util.dev()
        self.opt.zero_grad()
        n = batch.shape[0]
        for i in range(0, n, self.microbatch):
            micro = batch[i:i + self.microbatch].to(dev, non_blocking=True)
            micro_cond = {k: v[i:i + self.microbatch].to(dev, non_blocking=True) for k, v in cond.items()}
            last = (i + self.microbatch) >= n
            t, weights = self.schedule_sampler.sample(micro.shape[0], dev)
            self.buckets.enabled = last                      # no_sync semantics: reduce only with the last microbatch
            losses = self.diffusion.training_losses(self.model, micro, t, model_kwargs=micro_cond, rep_cond=self.rep_cond,
                                                    causal_modeling=self.causal_modeling)
            if isinstance(self.schedule_sampler, LossAwareSampler):
                self.schedule_sampler.update_with_local_losses(t, losses["loss"].detach())
            loss = (losses["loss"] * weights).mean()
            self.last_losses, self.last_t, self.last_w = {k: v.detach() for k, v in losses.items()}, t, weights
            # microbatches contribute mean losses of their own slice; scale like the reference (no extra scaling)
            loss.backward()
        self.buckets.finish(average=False)           # the SUM over ranks stays in the buffer; the optimizer kernel reads it times 1 / world
        self._grad_scale = 1.0 / self.world

    def optimize_normal(self):
        self._anneal_lr()
        self.opt.step(self._lr, grad_scale=getattr(self, "_grad_scale", 1.0))
        self._throttle()

    def _throttle(self):
        """Bounded run-ahead: the host may enqueue at most `run_ahead` optimizer steps beyond the one the GPU is executing.  Without a bound
        a GPU-bound loop fills the HIP launch queue and every further launch BUSY-WAITS for a slot (measured: 37 ms of CPU per 30 ms step
        of which 22 ms are work); the wait here is a blocking-sync event (the thread sleeps), so an 8-rank node keeps its cores for the
        launch threads and RCCL's proxies."""
        if self.run_ahead is None or dist_util.dev().type != "cuda":
            return
        ev = th.cuda.Event()
        ev.record()
        self._step_events.append(ev)
        if len(self._step_events) > self.run_ahead:
            old = self._step_events.pop(0)
            if self.throttle_poll_s > 0:
                while not old.query():              # (hipEventSynchronize spins, blocking-sync flag or not: measured 43 vs 38 ms of CPU per step)
                    time.sleep(self.throttle_poll_s)
            else:
                old.synchronize()

    def _anneal_lr(self):
        self._lr = self.lr
        if self.lr_anneal_steps:
            self._lr = self.lr * (1 - (self.step + self.resume_step) / self.lr_anneal_steps)

    def log_step(self):
        logger.logkv("step", self.step + self.resume_step)
        logger.logkv("samples", (self.step + self.resume_step + 1) * self.global_batch)
        if self.step % self.log_interval == 0 and self.last_losses is not None:      # one host sync per log interval
            log_loss_dict(self.diffusion, self.last_t, {k: v * self.last_w for k, v in self.last_losses.items()})
            logger.logkv_mean("grad_norm", float(np.sqrt(self.opt.grad_sqsum(getattr(self, "_grad_scale", 1.0)))))
            self.range_guard("training step")    # an operand left the f16 range of the split-precision planes: stop instead of training on NaNs

    def range_guard(self, what):
        """The library's range flag (include/cdae.h cdae_range_status), agreed on by ALL ranks: the flag is per process, and a rank that
        raised alone would leave its peers waiting in the next bucket all-reduce until the RCCL timeout."""
        if dist_util.dev().type != "cuda":
            return
        from ._lib import CdaeRangeError, range_check
        bad = 0
        try:
            range_check(what)
        except CdaeRangeError as e:
            bad, err = 1, e
        if self.world > 1:
            backend = dist.get_backend()
            flag = th.tensor([bad], dtype=th.int32, device=dist_util.dev() if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) and not bad:
                raise CdaeRangeError(f"{what}: another rank reported a non-finite contraction result (f16 split-precision range); stopping with it")
        if bad:
            raise err

    def _load_resume_state(self, resume_checkpoint):
        """EMA per rate and the Adam moments of the checkpoint's step.  `ema_<rate>_<step>.pt` / `opt<step>.pt` (upstream
        improved-diffusion's names) when this trainer wrote them; else the reference's single `ema_checkpoint.pt`, which its loop
        overwrites once per rate (train_util.py:319-345) and which therefore holds the LAST rate."""
        d, step = os.path.dirname(resume_checkpoint), self.resume_step
        f = self.opt.flat

        def into(flat_buf, sd, partial=False):
            for n, p, o in zip(f.names, f.params, f.offsets):
                if partial and n not in sd:
                    continue
                flat_buf.as_strided(p.shape, p.stride(), o).copy_(sd[n])

        found = set()
        for i, rate in enumerate(self.ema_rate):
            path = os.path.join(d, f"ema_{rate}_{step:06d}.pt")
            if os.path.exists(path):
                into(self.opt.ema[i], dist_util.load_state_dict(path, map_location="cpu"))
                found.add(i)
        legacy = os.path.join(d, "ema_checkpoint.pt")
        if (len(self.ema_rate) - 1) not in found and os.path.exists(legacy):
            into(self.opt.ema[-1], dist_util.load_state_dict(legacy, map_location="cpu"))
        opt_path = os.path.join(d, f"opt{step:06d}.pt")
        if os.path.exists(opt_path):
            st = dist_util.load_state_dict(opt_path, map_location="cpu")
            if "state" in st and "param_groups" in st:
                # torch.optim.AdamW.state_dict() — what the reference writes and reads under this name (train_util.py:159-169, 340-343) and
                # what save() below writes: per-parameter state indexed in model.parameters() order
                order = [n for n, _ in self.model.named_parameters()]
                idx = [i for g in st["param_groups"] for i in g["params"]]
                if len(idx) != len(order):
                    logger.log(f"{opt_path}: optimizer state for {len(idx)} parameters, the model has {len(order)}; Adam moments start from zero")
                    return
                state = st["state"]
                into(self.opt.m, {n: state[i]["exp_avg"] for n, i in zip(order, idx) if i in state}, partial=True)
                into(self.opt.v, {n: state[i]["exp_avg_sq"] for n, i in zip(order, idx) if i in state}, partial=True)
                steps = [int(float(state[i]["step"])) for i in idx if i in state]
                self.opt.t = max(steps) if steps else 0
            elif "exp_avg" in st and "exp_avg_sq" in st:          # round-1/2 private layout of this trainer
                into(self.opt.m, st["exp_avg"])
                into(self.opt.v, st["exp_avg_sq"])
                self.opt.t = int(st["step"])
            else:
                logger.log(f"{opt_path}: unknown optimizer checkpoint layout (keys {sorted(st)[:4]}); Adam moments start from zero")

    # ------------------------------------------------------------------ checkpoints (names of train_util.py:319-345)
    def save(self):
        self.range_guard("checkpoint")          # never write weights that consumed non-finite gradients
        if self.rank == 0:             # the reference writes on rank 1 only (so never in single-process runs): fixed, SURVEY Q6
            d = get_blob_logdir()
            if d:
                os.makedirs(d, exist_ok=True)
                sd = {k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()}
                th.save(sd, os.path.join(d, f"model{(self.step + self.resume_step):06d}.pt"))
                step = self.step + self.resume_step
                for i, rate in enumerate(self.ema_rate):
                    esd = {k: v.detach().cpu().contiguous() for k, v in self.opt.ema_state_dict(i).items()}
                    th.save(esd, os.path.join(d, "ema_checkpoint.pt"))          # the reference's name (one file, the last rate wins)
                    th.save(esd, os.path.join(d, f"ema_{rate}_{step:06d}.pt"))    # + one file per rate, so a resume restores every rate
                th.save(self.opt.torch_state_dict(), os.path.join(d, f"opt{step:06d}.pt"))
        if dist.is_initialized():
            dist.barrier()


def parse_resume_step_from_filename(filename):
    """path/to/modelNNNNNN.pt -> NNNNNN (0 if it does not parse), reference train_util.py:366-378."""
    parts = filename.split("model")
    if len(parts) < 2:
        return 0
    try:
        return int(parts[-1].split(".")[0])
    except ValueError:
        return 0


def get_blob_logdir():
    return os.environ.get("DIFFUSION_BLOB_LOGDIR", logger.get_dir())


def find_resume_checkpoint():
    return None


def log_loss_dict(diffusion, ts, losses):
    """Mean of every loss term plus its mean per timestep quartile (`<key>_q0..3`), the keys of train_util.py:401-407.
    Called once per log interval here (the reference calls it — and syncs the device — on every microbatch)."""
    ts = ts.detach().cpu().numpy().reshape(-1)
    for key, values in losses.items():
        v = values.detach().float().cpu().numpy()
        logger.logkv_mean(key, float(v.mean()))
        if v.ndim == 0 or v.shape[0] != ts.shape[0]:      # scalar term (masked kld_rep): no per-sample split
            continue
        for sub_t, sub_loss in zip(ts, v.reshape(ts.shape[0], -1).mean(axis=1)):
            logger.logkv_mean(f"{key}_q{int(4 * sub_t / diffusion.num_timesteps)}", float(sub_loss))
