"""Building blocks with the reference's names, constructor signatures and state-dict keys
(reference improved_diffusion/nn.py), computing through the HIP kernels in ops.py.

Parameters are created and initialised on the host (torch is plumbing for memory + init RNG); every
forward runs on the GPU through libcdae.so and raises on CPU tensors.
"""
import math

import torch as th
import torch.nn as nn

from . import ops

REP_EPS = 1e-8


# ----------------------------------------------------------------------------- primitive layers
class SiLU(nn.Module):
    """x * sigmoid(x)  (reference nn.py:430-432)."""

    def forward(self, x):
        return ops.silu(x)


class LeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.01):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, x):       # only reached when someone runs the Sequential by hand
        raise NotImplementedError("LeakyReLU is fused into the preceding BatchNorm / Linear kernel")


class Identity(nn.Identity):
    pass


class ReLU(nn.Module):
    def forward(self, x):       # only reached when someone runs the Sequential by hand
        raise NotImplementedError("ReLU is applied by the preceding Linear (act=ops.ACT_RELU)")


class Sigmoid(nn.Module):
    def forward(self, x):
        raise NotImplementedError("Sigmoid is applied by the preceding Linear (act=ops.ACT_SIGMOID)")


class _DropoutFn(th.autograd.Function):
    """y = x * mask / (1 - p) on the elementwise kernel; backward dx = dy * mask / (1 - p)."""

    @staticmethod
    def forward(ctx, x, mask, scale):
        x = x.contiguous() if not x.is_contiguous(memory_format=th.channels_last) else x
        y = th.empty_like(x)
        ops.check(ops.lib.cdae_mul_scale(ops.ptr(x), ops.ptr(mask), float(scale), ops.ptr(y), x.numel(), ops.stream()))
        ctx.save_for_backward(mask)
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous(memory_format=th.channels_last) if mask.is_contiguous(memory_format=th.channels_last) and not mask.is_contiguous() else dy.contiguous()
        dx = th.empty_like(dy)
        ops.check(ops.lib.cdae_mul_scale(ops.ptr(dy), ops.ptr(mask), ctx.scale, ops.ptr(dx), dy.numel(), ops.stream()))
        return dx, None, None


class Dropout(nn.Module):
    """nn.Dropout(p) of the ResBlock's out_layers (reference unet.py:153).  Identity in eval mode and for p = 0 (every CausalDiffAE
    config); in training the keep mask ~ Bernoulli(1 - p) is drawn on the device (or injected: rng_override(dropout_mask=...), one
    entry consumed per call) and applied by cdae_mul_scale.  The block then runs its unfused path (the fused ResBlock node has no
    mask input), like the reference runs dropout as its own module."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p <= 0.0 or not self.training:
            return x
        if not th.is_tensor(x):
            raise TypeError("Dropout needs a plain tensor (the ResBlock asks its GroupNorm for one when dropout is active)")
        inj = _RNG_OVERRIDE.get("dropout_mask")
        if inj is not None:
            mask = inj.pop(0) if isinstance(inj, list) else inj
            mask = mask.to(x.device).float().expand_as(x)
        else:
            mask = th.bernoulli(th.full(x.shape, 1.0 - self.p, device=x.device))
        # same memory order as x (NHWC storage of a logical NCHW tensor): the kernel multiplies storage element by storage element
        mask = mask.contiguous(memory_format=th.channels_last) if (x.dim() == 4 and x.is_contiguous(memory_format=th.channels_last) and not x.is_contiguous()) \
            else mask.contiguous()
        return _DropoutFn.apply(x, mask, 1.0 / (1.0 - self.p))


class Linear(nn.Module):
    """nn.Linear replacement: y = x W^T + b on the MFMA GEMM kernel."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(th.empty(out_features, in_features))
        self.bias = nn.Parameter(th.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, act=ops.ACT_NONE, res=None):
        return ops.linear(x, self.weight, self.bias, res=res, act=act)


class Embedding(nn.Module):
    def __init__(self, num, dim):
        super().__init__()
        self.weight = nn.Parameter(th.randn(num, dim))


class ConvNd(nn.Module):
    """3x3 (pad 1, stride 1|2) or 1x1 convolution.  The weight keeps the reference's logical shape
    ([Cout,Cin,k,k] / [Cout,Cin,1] for dims=1) so state dicts interchange; 3x3 weights are stored OHWI."""

    def __init__(self, dims, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        if dims not in (1, 2):
            raise ValueError(f"unsupported dimensions: {dims}")
        if not ((kernel_size == 3 and padding == 1 and dims == 2) or (kernel_size == 1 and padding == 0)):
            raise ValueError("only 3x3/pad 1 (2-D) and 1x1 convolutions exist on the CausalDiffAE hot path")
        self.dims, self.in_channels, self.out_channels = dims, in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        shape = (out_channels, in_channels) + (kernel_size,) * dims
        w = th.empty(shape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        if kernel_size == 3:
            w = w.contiguous(memory_format=th.channels_last)
        self.weight = nn.Parameter(w)
        bound = 1 / math.sqrt(in_channels * kernel_size ** dims)
        self.bias = nn.Parameter(th.empty(out_channels).uniform_(-bound, bound))

    def forward(self, x, res=None, up=False, out_nchw=False, emit_split=False, gn_stats=False):
        if isinstance(x, ops.LazyGN):          # GroupNorm output not written yet: fuse it into the conv where the kernel exists
            if self.kernel_size == 3 and out_nchw and res is None and not up and self.stride == 1 and not emit_split and ops.head_conv_ok(x, self.weight):
                return ops.head_conv(x, self.weight, self.bias)          # the output head: a few channels, exact fp32
            N_, C_, H_, W_ = x.shape
            x = x.planes(gm=self.kernel_size == 3 and self.stride == 1 and not up and not out_nchw
                         and ops.gm_wanted(N_, H_, W_, C_, self.weight.shape[0]))      # group-major for the window conv kernel (where this conv runs on it)
        if isinstance(x, ops.SplitAct):
            assert self.kernel_size == 3
            return ops.conv3x3_ps(x, self.weight, self.bias, res=res, stride=self.stride, up=up, out_nchw=out_nchw, emit_split=emit_split,
                                  gn_stats=gn_stats)
        if isinstance(x, ops.CatAct):
            if self.kernel_size == 1 and res is None:
                return ops.conv1x1_cat(x, self.weight, self.bias)
            x = ops.materialize(x)
        if self.kernel_size == 3:
            if self.stride == 1 and res is None and not up and not out_nchw and ops.presplit_ok() and ops.stem_conv_gn_ok(x, self.weight):
                return ops.stem_conv_gn(x, self.weight, self.bias)          # the network's input conv: its result carries the GroupNorm sums
            return ops.conv3x3(x, self.weight, self.bias, res=res, stride=self.stride, up=up, out_nchw=out_nchw)
        if x.dim() == 3:       # [B, C, T] (AttentionBlock convention)
            y = ops.conv1x1(x.unsqueeze(-1), self.weight, self.bias, None if res is None else res.unsqueeze(-1))
            return y.squeeze(-1)
        return ops.conv1x1(x, self.weight, self.bias, res)


def conv_nd(dims, *args, **kwargs):
    """Create a 1D or 2D convolution module (reference nn.py:470-480)."""
    return ConvNd(dims, *args, **kwargs)


def linear(*args, **kwargs):
    return Linear(*args, **kwargs)


class AvgPool2(nn.Module):
    """nn.AvgPool2d(kernel_size=2, stride=2) on the library's kernels (even image sizes)"""

    def forward(self, x):
        return ops.avg_pool2(x)


def avg_pool_nd(dims, *args, **kwargs):
    """Create a 2D average pooling module (reference nn.py:483-493; the UNet asks for kernel_size = stride = 2, unet.py:101-103)."""
    if dims != 2 or (args and args[0] != 2) or kwargs.get("kernel_size", 2) != 2 or kwargs.get("stride", 2) not in (2, None):
        raise NotImplementedError("avg_pool_nd: 2-D pooling with kernel_size = stride = 2 (what the UNet builds) only")
    return AvgPool2()


class GroupNorm32(nn.Module):
    """GroupNorm with fp32 statistics (reference nn.py:435-437)."""

    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(th.ones(num_channels))
        self.bias = nn.Parameter(th.zeros(num_channels))

    def forward(self, x, scale_shift=None, silu=False, split=False, coef=False):
        """split=True: the caller feeds the result straight into a conv3x3 / 1x1 GEMM; in no-grad f16 modes it is then written
        as pre-split f16 planes (ops.SplitAct) for the LDS-DMA kernel.  coef=True: the consumer is likely to apply the norm itself from
        the per-(image, channel) coefficient table (ops.LazyGN.coefficients) — the statistics launch writes that table as well."""
        if split and len(x.shape) == 4 and ops.presplit_ok() and ops.can_split(self.num_channels, self.num_groups):
            return ops.group_norm_lazy(x, self.weight, self.bias, scale_shift, silu, self.num_groups, self.eps, want_coef=coef)
        if scale_shift is not None and not scale_shift.is_contiguous():          # a column slice of the batched emb_layers GEMM
            scale_shift = scale_shift.contiguous()
        return ops.group_norm(ops.materialize(x), self.weight, self.bias, scale_shift, silu, self.num_groups, self.eps)


def normalization(channels):
    return GroupNorm32(32, channels)


class BatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(th.ones(num_features))
        self.bias = nn.Parameter(th.zeros(num_features))
        self.register_buffer("running_mean", th.zeros(num_features))
        self.register_buffer("running_var", th.ones(num_features))
        self.register_buffer("num_batches_tracked", th.tensor(0, dtype=th.long))

    def forward(self, x, slope=0.01):
        """BatchNorm fused with the LeakyReLU that always follows it in the encoder."""
        if self.training:
            self.num_batches_tracked += 1
        return ops.bn_lrelu(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                            self.eps, self.momentum, slope)


# ----------------------------------------------------------------------------- helpers kept from the reference API
def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def scale_module(module, scale):
    for p in module.parameters():
        p.detach().mul_(scale)
    return module


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


def update_ema(target_params, source_params, rate=0.99):
    for targ, src in zip(target_params, source_params):
        targ.detach().mul_(rate).add_(src, alpha=1 - rate)


_FREQ_CACHE = {}


def _freqs(dim, max_period, device):
    key = (dim, max_period, str(device))
    f = _FREQ_CACHE.get(key)
    if f is None:
        half = dim // 2
        # computed on the host exactly like the reference (nn.py:562-564) and uploaded once
        f = th.exp(-math.log(max_period) * th.arange(start=0, end=half, dtype=th.float32) / half).to(device)
        _FREQ_CACHE[key] = f
    return f


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal timestep embeddings [N x dim] (reference nn.py:551-569)."""
    return ops.timestep_embedding(timesteps, _freqs(dim, max_period, timesteps.device), dim)


def checkpoint(func, inputs, params, flag):
    """Activation recomputation (reference nn.py:572-618) via torch.utils.checkpoint."""
    if flag:
        from torch.utils.checkpoint import checkpoint as _ckpt
        return _ckpt(func, *inputs, use_reentrant=False)
    return func(*inputs)


def kl_normal(qm, qv, pm, pv):
    """Element-wise KL(N(qm,qv) || N(pm,pv)) summed over the last dim (reference nn.py:440-457)."""
    element_wise = 0.5 * (th.log(pv) - th.log(qv) + qv / pv + (qm - pm).pow(2) / pv - 1)
    return element_wise.sum(-1)


_RNG_OVERRIDE = {}


class rng_override:
    """Context manager that injects the normal / Bernoulli draws of UNetModel.forward (tests, parity):
    with rng_override(eps_z=..., cfg_mask=...): model(...)"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        _RNG_OVERRIDE.update(self.kw)

    def __exit__(self, *a):
        for k in self.kw:
            _RNG_OVERRIDE.pop(k, None)


def reparameterize(m, v, eps=None):
    """z = m + sqrt(v) * eps (reference nn.py:460-467).  eps is drawn on the device unless injected."""
    if eps is None:
        eps = _RNG_OVERRIDE.get("eps_z")
    if eps is None:
        eps = th.randn(m.size(), device=m.device)
    return m + (v ** 0.5) * eps.to(m.device)


# ----------------------------------------------------------------------------- causal normalising flow (flow_based=True)
class MultivariateCausalFlow(nn.Module):
    """Affine autoregressive flow over `dim` latent blocks of width `k` (reference nn.py:342-426).  Block i is transformed
    as z_i = exp(s_i) e_i + t_i with (s_i, t_i) = conditioner(z * column i of C repeated k times); the conditioners are two
    3-layer MLPs (ReLU, ReLU, sigmoid) shared by all blocks, run on the GEMM kernels.  Parameter names match the reference
    (`s_cond.{0,2,4}`, `t_cond.{0,2,4}`).  The reference reshapes to (-1, 2, 256) whatever dim/k say; here dim/k are used.

    Reference behaviours kept on purpose: `flow` conditions block i on the blocks already produced (later blocks, block i
    itself included, are still zero), while `reverse` conditions on the FULL z — so `reverse` is not the inverse of `flow`
    whenever C has a non-zero diagonal, as it does for the caller's C = I - A (unet.py:581)."""

    def __init__(self, dim, k, nh=100):
        super().__init__()
        self.dim, self.k = dim, k
        self.s_cond = nn.Sequential(Linear(dim * k, nh), ReLU(), Linear(nh, nh), ReLU(), Linear(nh, k), Sigmoid())
        self.t_cond = nn.Sequential(Linear(dim * k, nh), ReLU(), Linear(nh, nh), ReLU(), Linear(nh, k), Sigmoid())

    @staticmethod
    def _run(net, x):
        return net[4](net[2](net[0](x, act=ops.ACT_RELU), act=ops.ACT_RELU), act=ops.ACT_SIGMOID)

    def _column_mask(self, C, i, device):
        col = th.as_tensor(C, dtype=th.float32)[:, i].to(device)
        if not bool((col == 1).any()):                 # "no parents" branch of the reference
            col = th.zeros_like(col)
        return col.repeat_interleave(self.k)           # == C[:, i].repeat(k, 1).T.reshape(dim * k)

    def flow(self, e, C):
        """e [N, dim*k] -> [z [N, dim*k], log_det [N]] with log_det = sum of all slopes."""
        N = e.shape[0]
        e3 = e.reshape(N, self.dim, self.k)
        blocks = []
        log_det = th.zeros(N, device=e.device)
        for i in range(self.dim):
            done = blocks + [th.zeros(N, self.k, device=e.device)] * (self.dim - len(blocks))
            inp = th.cat(done, dim=1) * self._column_mask(C, i, e.device)
            s, t = self._run(self.s_cond, inp), self._run(self.t_cond, inp)
            blocks.append(th.exp(s) * e3[:, i, :] + t)
            log_det = log_det + s.sum(dim=1)
        return [th.cat(blocks, dim=1), log_det]

    def reverse(self, z, C):
        """z [N, dim*k] -> [log_det [N] (minus the slopes), log N(e; 1, I)] — see the class note on what this inverts."""
        N = z.shape[0]
        z3 = z.reshape(N, self.dim, self.k)
        log_det = th.zeros(N, device=z.device)
        es = []
        for i in range(self.dim):
            inp = z.reshape(N, -1) * self._column_mask(C, i, z.device)
            s, t = self._run(self.s_cond, inp), self._run(self.t_cond, inp)
            es.append(th.exp(-s) * (z3[:, i, :] - t))
            log_det = log_det - s.sum(dim=1)
        e = th.cat(es, dim=1)
        log_prob = -0.5 * ((e - 1.0) ** 2).sum(dim=1) - 0.5 * e.shape[1] * math.log(2 * math.pi)     # MultivariateNormal(ones, I)
        return [log_det, log_prob]


# ----------------------------------------------------------------------------- causal semantic encoder
class GaussianConvEncoder(nn.Module):
    """[conv3x3 s2 -> BatchNorm2d -> LeakyReLU] x L -> flatten -> (fc_mu, softplus(fc_var)+1e-8)
    (reference nn.py:15-110).  Constructor signature and parameter names follow the reference."""

    def __init__(self, in_channels, latent_dim, hidden_dims=None, beta=4, gamma=1000., max_capacity=25,
                 Capacity_max_iter=1e5, loss_type='B', num_vars=4, **kwargs):
        super().__init__()
        self.latent_dim, self.in_channels, self.num_vars = latent_dim, in_channels, num_vars
        if hidden_dims is None:
            hidden_dims = [16, 32, 32, 64, 64, 128] if num_vars == 4 else [16, 32, 64, 128]
        self.hidden_dims = list(hidden_dims)
        mods = []
        c = in_channels
        for h in hidden_dims:
            mods.append(nn.Sequential(ConvNd(2, c, h, 3, stride=2, padding=1), BatchNorm2d(h), LeakyReLU()))
            c = h
        self.encoder = nn.Sequential(*mods)
        self.fc_mu = Linear(hidden_dims[-1] * 4, latent_dim)
        self.fc_var = Linear(hidden_dims[-1] * 4, latent_dim)

    def encode(self, input):
        h = input
        for blk in self.encoder:
            h = blk[1](blk[0](h))                       # conv -> fused BN+LeakyReLU
        flat = ops.to_nchw(h).reshape(h.shape[0], -1)   # th.flatten on the NCHW order the fc weights expect
        mu = self.fc_mu(flat)
        var = ops.softplus_eps(self.fc_var(flat), REP_EPS)
        return [mu, var]


class GaussianConvEncoderClf(GaussianConvEncoder):
    """The evaluation classifier / regressor of the reference's test scripts (reference nn.py:115-222, built by
    scripts/image_causaldae_test.py:131-156 to score counterfactuals): the same strided conv encoder plus one Linear head on the
    flattened features; forward(x) -> [N, 1].  Same parameter names (encoder.*, fc_mu, fc_var, fc), so its checkpoints load."""

    def __init__(self, in_channels, latent_dim, hidden_dims=None, num_vars=4, **kwargs):
        super().__init__(in_channels, latent_dim, hidden_dims=hidden_dims, num_vars=num_vars, **kwargs)
        self.fc = Linear(self.hidden_dims[-1] * 4, 1)

    def forward(self, x):
        h = x
        for blk in self.encoder:
            h = blk[1](blk[0](h))
        return self.fc(ops.to_nchw(h).reshape(h.shape[0], -1))


class MLP(nn.Module):
    """Linear(d -> latent) LeakyReLU Linear(latent -> d)  (reference nn.py:225-240)."""

    def __init__(self, latent_dim, num_var):
        super().__init__()
        self.latent_dim, self.num_var = latent_dim, num_var
        self.net = nn.Sequential(Linear(latent_dim // num_var, latent_dim), LeakyReLU(), Linear(latent_dim, latent_dim // num_var))

    def forward(self, x, res=None):
        return self.net[2](self.net[0](x, act=ops.ACT_LRELU), res=res)


class CausalModeling(nn.Module):
    """Mask by the adjacency A, per-variable MLP, add back the exogenous noise (reference nn.py:244-312)."""

    def __init__(self, latent_dim, num_var=None, learn=False, **kwargs):
        super().__init__()
        self.latent_dim, self.num_var = latent_dim, num_var
        if learn:
            self.A = nn.Parameter(th.zeros(num_var, num_var))
        else:
            self.A = th.tensor([[0, 1], [0, 0]])
        self.nonlinearities = nn.ModuleDict({str(i): MLP(latent_dim=latent_dim, num_var=num_var) for i in range(num_var)})

    def causal_masking(self, u, A):
        return ops.causal_mask(u, A.to(u.device).float(), self.num_var)            # A^T u on [N, nv, d]

    def nonlinearity_add_back_noise(self, u, z_pre):
        N = u.shape[0]
        d = self.latent_dim // self.num_var
        # (unbind, not z_pre[:, i]: one UnbindBackward node = one stack kernel per tensor in the backward, where every SelectBackward is a
        #  zeros + copy + add on the full tensor)
        zs = z_pre.unbind(1)
        us = u.reshape(N, self.num_var, d).unbind(1)
        outs = [self.nonlinearities[str(i)](zs[i], res=us[i].contiguous()) for i in range(self.num_var)]
        return th.cat(outs, dim=1)
