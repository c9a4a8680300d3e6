"""GPU parity of each HIP kernel (through the C-ABI wrappers in causaldiffae_amd.ops) against a plain
torch-CPU fp32 restatement of the same op / the oracle.  Tolerances are stated per test; the bar for the
whole path is fp32 within 1e-4 (BASELINE north_star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def err(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def precision_scope_f16x3():
    from causaldiffae_amd._lib import precision_scope
    return precision_scope("f16x3")


def chlast_param(w):
    return w.contiguous(memory_format=torch.channels_last).to(DEV)


# ------------------------------------------------------------------ GEMM / linear
@pytest.mark.parametrize("M,N,K", [(16, 512, 128), (2, 1024, 512), (300, 200, 68), (1024, 1152, 384), (4096, 128, 256),
                                   (65, 33, 4), (8192, 256, 512)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear(M, N, K, act, precision):
    from causaldiffae_amd import ops
    x, w, b, r = rnd(M, K), rnd(N, K, seed=1) / K ** 0.5, rnd(N, seed=2), rnd(M, N, seed=3)
    y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), res=r.to(DEV), act=act)
    ref = F.linear(x.double(), w.double(), b.double()) + r.double()
    ref = [ref, ref * torch.sigmoid(ref), F.leaky_relu(ref, 0.01)][act]
    assert err(y, ref) < 2e-5


@pytest.mark.parametrize("M,N,K,bias,res", [(4096, 384, 384, True, True), (4100, 200, 96, True, False), (8192, 512, 512, False, True),
                                            (32768, 128, 1024, False, False), (5000, 64, 32, True, True)])
def test_linear_stream_kernel(M, N, K, bias, res):
    """cdae_linear_fwd_stream (large-M linears / 1x1 convs in f16x3 mode: fp32 rows streamed once, pre-split weight planes) vs fp64 and
    vs the igemm-loader kernel it replaces; partial row and column tiles, with and without bias / residual; a 1x1 conv weight view."""
    from causaldiffae_amd import ops
    x, w = rnd(M, K).to(DEV), (rnd(N, K, seed=1) / K ** 0.5).to(DEV)
    b = rnd(N, seed=2).to(DEV) if bias else None
    r = rnd(M, N, seed=3).to(DEV) if res else None
    with precision_scope_f16x3():
        assert ops._stream_gemm_ok(x, M, N, K, 0, 1.0, r)
        y = ops.linear(x, w, b, res=r)
        y4 = ops.linear(x, w.reshape(N, K, 1, 1), b, res=r)              # the [Cout, Cin, 1, 1] form of a 1x1 conv weight
        saved, ops._STREAM_GEMM = ops._STREAM_GEMM, False
        try:
            old = ops.linear(x, w, b, res=r)
        finally:
            ops._STREAM_GEMM = saved
    ref = F.linear(x.double(), w.double(), None if b is None else b.double()) + (0 if r is None else r.double())
    assert torch.equal(y, y4)
    assert err(y, ref) < 4e-6 * ref.abs().max().item()
    assert err(y, old) < 4e-6 * ref.abs().max().item()
    if M % 64 == 0 and N % 4 == 0:          # the same GEMM with its result also leaving as f16 planes (attention proj_out ahead of a resample conv)
        with precision_scope_f16x3(), torch.no_grad():
            y2, planes = ops.linear_emit(x, w, b, r, (M // 64, N, 8, 8))
        assert torch.equal(y2, y)
        hi = y.half()
        assert torch.equal(planes.hi.reshape(-1, N), hi) and torch.equal(planes.lo.reshape(-1, N), (y - hi.float()).half())


@pytest.mark.parametrize("N,C1,C2,Cout,S", [(128, 128, 0, 128, 32), (32, 256, 128, 256, 32), (8, 128, 128, 128, 64), (96, 512, 384, 512, 8), (4, 128, 0, 128, 16)])
def test_group_major_planes(N, C1, C2, Cout, S):
    """Group-major activation planes ([C / 16][pixels][16]: the window conv kernel's contiguous half-windows): written by the GroupNorm apply
    kernel and by the entry sweep, they hold exactly the pixel-major planes' values (cdae_planes_gm_to_pc), and the conv on them equals the
    conv on the pixel-major planes bit for bit — on the window kernel, on split-K tiles and where the shape falls back to another kernel."""
    from causaldiffae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(51)
    a = ops.to_nhwc(torch.randn(N, C1, S, S, device=dev, generator=g))
    x = ops.CatAct(a, ops.to_nhwc(torch.randn(N, C2, S, S, device=dev, generator=g))) if C2 else a
    C = C1 + C2
    gamma, beta = 1 + 0.1 * torch.randn(C, device=dev, generator=g), 0.1 * torch.randn(C, device=dev, generator=g)
    w = (torch.randn(Cout, C, 3, 3, device=dev, generator=g) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
    b = 0.1 * torch.randn(Cout, device=dev, generator=g)
    res = ops.to_nhwc(torch.randn(N, Cout, S, S, device=dev, generator=g))
    with torch.no_grad():
        lz = ops.group_norm_lazy(x, gamma, beta, None, True, 32, 1e-5)
        pc, gm = lz.planes(), lz.planes(gm=True)
        assert gm.gm and not pc.gm
        back = gm.pc()
        assert torch.equal(back.hi, pc.hi) and torch.equal(back.lo, pc.lo)
        y_pc = ops.conv3x3_ps(pc, w, b, res=res, gn_stats=True)
        y_gm = ops.conv3x3_ps(gm, w, b, res=res, gn_stats=True)
        assert torch.equal(y_pc, y_gm)
        if hasattr(y_pc, "_gnparts"):
            assert torch.equal(y_pc._gnparts, y_gm._gnparts)
        assert torch.equal(ops.conv3x3_ps(gm, w, b, stride=2), ops.conv3x3_ps(pc, w, b, stride=2))       # converted on the way
        if C2 or C != Cout:
            ws = torch.randn(Cout, C, 1, 1, device=dev, generator=g) / C ** 0.5
            if ops.skip_gn_ok(lz, ws):
                s1, p1 = ops.skip_gn_fused(lz, ws, b)
                s2, p2 = ops.skip_gn_fused(lz, ws, b, gm=True)
                assert p2.gm and torch.equal(s1, s2)
                q = p2.pc()
                assert torch.equal(q.hi, p1.hi) and torch.equal(q.lo, p1.lo)


def test_linear_backward():
    from causaldiffae_amd import ops
    M, N, K = 260, 384, 512
    x, w, b = rnd(M, K), rnd(N, K, seed=1) / K ** 0.5, rnd(N, seed=2)
    gy = rnd(M, N, seed=5)
    xd, wd, bd = [t.to(DEV).requires_grad_(True) for t in (x, w, b)]
    y = ops.linear(xd, wd, bd, act=1)
    (y * gy.to(DEV)).sum().backward()
    xc, wc, bc = [t.double().requires_grad_(True) for t in (x, w, b)]
    yc = F.silu(F.linear(xc, wc, bc))
    (yc * gy.double()).sum().backward()
    assert err(y, yc) < 2e-5
    assert err(xd.grad, xc.grad) < 5e-5
    assert err(wd.grad, wc.grad) < 2e-4          # sums over M=260 rows
    assert err(bd.grad, bc.grad) < 2e-4


@pytest.mark.parametrize("shapes,rows,grouped", [
    ([(384, 384)] * 5 + [(1152, 384)] * 3, 2048, True),          # the 16 x 16 level's mix: 64 x 64 tiles
    ([(512, 512)] * 9 + [(1536, 512)] * 5 + [(512, 1024)], 512, True),    # the 8 x 8 level's: 128 x 128 tiles
    ([(128, 128)] * 3, 4096, False),                               # too few tiles together: one launch each, K split
    ([(384, 380), (384, 384)] * 6, 1024, True),                    # ragged tile edges
])
def test_linear_wgrad_group(shapes, rows, grouped, precision):
    """cdae_linear_wgrad_group: the weight gradients of a level's 1 x 1 convs / linears (reference unet.py:165-171, 216-236 through
    autograd) in ONE unsplit launch where together they fill the chip — dw += dy^T x and dbias += column sums of dy, accumulating into
    buffers that already hold values — against float64; the launch log says whether the group form ran."""
    import ctypes
    from causaldiffae_amd import _lib
    from causaldiffae_amd._lib import LwItem, check, lib, ptr, splitk_ws, stream, SPLITK_BYTES
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(len(shapes) + rows)
    items, keep, want = [], [], []
    for i, (N, K) in enumerate(shapes):
        x = torch.randn(rows, K, generator=g)
        dy = torch.randn(rows, N, generator=g) * (0.5 + i % 3)
        dw0, db0 = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
        has_b = i % 2 == 0
        xd, dyd, dwd, dbd = x.to(dev), dy.to(dev), dw0.to(dev), db0.to(dev)
        keep.append((xd, dyd, dwd, dbd))
        items.append(LwItem(ptr(xd), ptr(dyd), ptr(dwd), ptr(dbd) if has_b else None, K, N, K, rows, N, K, 1))
        want.append((dw0.double() + dy.double().t() @ x.double(), db0.double() + (dy.double().sum(0) if has_b else 0.0)))
    arr = (LwItem * len(items))(*items)
    ws = splitk_ws(dev)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    _lib.prof_read()
    check(lib.cdae_linear_wgrad_group(arr, len(items), ptr(ws), SPLITK_BYTES, stream()))
    launches = _lib.prof_read()["igemm"]["launches"]
    _lib.prof_enable(False)
    assert (launches == 1) == grouped, launches          # (one by one: a launch per member)
    tol = 2e-5 if precision == "fp32" else 2e-4
    for (xd, dyd, dwd, dbd), (dw, db) in zip(keep, want):
        assert err(dwd, dw) < tol * float(dw.abs().max()), (tuple(dwd.shape), err(dwd, dw), float(dw.abs().max()))
        assert err(dbd, db) < tol * float(db.abs().max())


# ------------------------------------------------------------------ resampling without a conv (conv_resample=False)
def test_plain_resample_golden(golden):
    """Upsample(use_conv=False) (reference unet.py:51-78: nearest 2x alone) against the reference's own output and input gradient (G22);
    Downsample(use_conv=False) against torch's avg_pool2d — the reference module cannot be built (its avg_pool_nd(stride) call lacks
    `dims`: TypeError, recorded in the fixture), the docstring's meaning (2 x 2 average pool) is what is implemented; bf16 rows too."""
    from causaldiffae_amd.unet import Downsample, Upsample
    from oracle.closed_form import synth
    g = golden("g22_plain_resample.npz")
    assert int(g["down/reference_builds"]) == 0
    x = synth("T28r.x", (2, 64, 7, 14))
    gy = synth("T28r.gy", (2, 64, 14, 28))
    xd = x.to(DEV).requires_grad_(True)
    y = Upsample(64, False)(xd)
    (y * gy.to(DEV)).sum().backward()
    assert err(y, torch.from_numpy(g["up/y"])) == 0.0 and err(xd.grad, torch.from_numpy(g["up/dx"])) < 1e-6
    # average pool (and its gradient) against torch, odd channel count included (scalar kernels)
    for C, H, W in ((64, 14, 28), (6, 8, 8)):
        a = rnd(3, C, H, W)
        ga = rnd(3, C, H // 2, W // 2, seed=1)
        ad = a.to(DEV).requires_grad_(True)
        o = Downsample(C, False)(ad)
        (o * ga.to(DEV)).sum().backward()
        ac = a.double().requires_grad_(True)
        oc = F.avg_pool2d(ac, 2)
        (oc * ga.double()).sum().backward()
        assert err(o, oc) < 1e-6 and err(ad.grad, ac.grad) < 1e-6
        u = Upsample(C, False)(ad.detach())
        assert err(u, F.interpolate(a, scale_factor=2, mode="nearest")) == 0.0
    with pytest.raises(NotImplementedError):
        Downsample(8, False)(torch.zeros(1, 8, 7, 7, device=DEV))


# ------------------------------------------------------------------ conv3x3 forward
CONVS = [
    # N, Cin, Cout, H, stride, up, nchw_in, out_nchw
    (2, 128, 128, 16, 1, False, False, False),
    (2, 128, 256, 8, 1, False, False, False),
    (3, 384, 128, 8, 1, False, False, False),      # Cout < tile, Cin = 3*128
    (2, 256, 256, 16, 2, False, False, False),     # Downsample
    (2, 128, 128, 8, 1, True, False, False),       # Upsample fused
    (2, 4, 128, 16, 1, False, True, False),        # stem, NCHW input, generic gather
    (2, 1, 128, 12, 1, False, True, False),
    (2, 3, 128, 9, 1, False, True, False),         # odd size
    (2, 128, 4, 16, 1, False, False, True),        # output head -> NCHW
    (4, 4, 16, 32, 2, False, True, False),         # encoder conv (stride 2, tiny channels)
    (4, 16, 32, 16, 2, False, False, False),
    (4, 32, 64, 8, 2, False, False, False),
    (1, 1024, 512, 8, 1, False, False, False),     # deep K -> split-K
    (2, 640, 384, 16, 1, False, False, False),
]


@pytest.mark.parametrize("N,Cin,Cout,H,stride,up,nchw_in,out_nchw", CONVS)
def test_conv3x3_forward(N, Cin, Cout, H, stride, up, nchw_in, out_nchw, precision):
    from causaldiffae_amd import ops
    x, w, b = rnd(N, Cin, H, H), rnd(Cout, Cin, 3, 3, seed=1) / (9 * Cin) ** 0.5, rnd(Cout, seed=2)
    xin = x.to(DEV) if nchw_in else ops.to_nhwc(x.to(DEV))
    xr = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up else x.double()
    ref = F.conv2d(xr, w.double(), b.double(), stride=stride, padding=1)
    res = None
    if not out_nchw:
        r = rnd(*ref.shape, seed=7)
        res = ops.to_nhwc(r.to(DEV))
        ref = ref + r.double()
    y = ops.conv3x3(xin, chlast_param(w), b.to(DEV), res=res, stride=stride, up=up, out_nchw=out_nchw)
    assert tuple(y.shape) == tuple(ref.shape)
    if out_nchw:
        assert y.is_contiguous()
    assert err(y, ref) < 3e-5


@pytest.mark.parametrize("N,Cin,Cout,H,stride,up,nchw_in", [
    (2, 128, 128, 8, 1, False, False), (2, 128, 256, 8, 1, False, False), (2, 256, 128, 8, 2, False, False),
    (2, 128, 128, 4, 1, True, False), (2, 4, 128, 8, 1, False, True), (2, 128, 4, 8, 1, False, False),
    (3, 16, 32, 8, 2, False, False), (2, 1, 16, 14, 2, False, True), (2, 384, 256, 8, 1, False, False)])
def test_conv3x3_backward(N, Cin, Cout, H, stride, up, nchw_in):
    from causaldiffae_amd import ops
    x, w, b = rnd(N, Cin, H, H), rnd(Cout, Cin, 3, 3, seed=1) / (9 * Cin) ** 0.5, rnd(Cout, seed=2)
    xd = (x.to(DEV) if nchw_in else ops.to_nhwc(x.to(DEV))).requires_grad_(True)
    wd = chlast_param(w).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    y = ops.conv3x3(xd, wd, bd, stride=stride, up=up)
    xc, wc, bc = [t.double().requires_grad_(True) for t in (x, w, b)]
    xr = F.interpolate(xc, scale_factor=2, mode="nearest") if up else xc
    yc = F.conv2d(xr, wc, bc, stride=stride, padding=1)
    gy = rnd(*yc.shape, seed=9)
    (y * gy.to(DEV)).sum().backward()
    (yc * gy.double()).sum().backward()
    assert err(y, yc) < 3e-5
    assert err(xd.grad, xc.grad) < 1e-4
    scale = max(1.0, wc.grad.abs().max().item())
    assert err(wd.grad, wc.grad) < 1e-4 * scale
    assert err(bd.grad, bc.grad) < 1e-4 * max(1.0, bc.grad.abs().max().item())
    assert wd.grad.permute(0, 2, 3, 1).is_contiguous() or Cin == 1


# ------------------------------------------------------------------ GroupNorm
@pytest.mark.parametrize("N,C,H", [(2, 128, 16), (2, 384, 8), (3, 640, 4), (2, 896, 8), (2, 1024, 8), (2, 128, 64), (2, 32, 7), (2, 96, 6), (9, 256, 16), (17, 128, 32)])
@pytest.mark.parametrize("ssn,silu", [(False, True), (True, True), (False, False)])
@pytest.mark.parametrize("fold2", [1, 0])
def test_group_norm(N, C, H, ssn, silu, fold2):
    """forward + backward against float64 autograd; fold2: the backward as two launches (group fold in the dx kernel's prologue, channel
    folds as extra rows of its grid — the default since round 6) and with the separate fold launch (cdae_tune_set)"""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import tune_scope
    with tune_scope(gn_bwd_fold2=fold2):
        _group_norm_case(ops, N, C, H, ssn, silu)


def _group_norm_case(ops, N, C, H, ssn, silu):
    x = rnd(N, C, H, H, lo=-2, hi=3)
    x = x + torch.linspace(-3, 3, C)[None, :, None, None]        # per-channel offsets: mean^2 >> var inside groups
    g, b = rnd(C, seed=1) + 1.5, rnd(C, seed=2)
    ss = rnd(N, 2 * C, seed=3) if ssn else None
    gd, bd = [t.to(DEV).requires_grad_(True) for t in (g, b)]
    xd = ops.to_nhwc(x.to(DEV)).requires_grad_(True)
    ssd = ss.to(DEV).requires_grad_(True) if ssn else None
    y = ops.group_norm(xd, gd, bd, ssd, silu)
    xc, gc, bc = [t.double().requires_grad_(True) for t in (x, g, b)]
    ssc = ss.double().requires_grad_(True) if ssn else None
    h = F.group_norm(xc, 32, gc, bc, 1e-5)
    if ssn:
        h = h * (1 + ssc[:, :C, None, None]) + ssc[:, C:, None, None]
    yc = F.silu(h) if silu else h
    gy = rnd(N, C, H, H, seed=4)
    (y * gy.to(DEV)).sum().backward()
    (yc * gy.double()).sum().backward()
    assert err(y, yc) < 2e-5
    assert err(xd.grad, xc.grad) < 1e-4
    assert err(gd.grad, gc.grad) < 1e-4 * max(1.0, gc.grad.abs().max().item())
    assert err(bd.grad, bc.grad) < 1e-4 * max(1.0, bc.grad.abs().max().item())
    if ssn:
        assert err(ssd.grad, ssc.grad) < 1e-4 * max(1.0, ssc.grad.abs().max().item())


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,T,heads,ch", [(2, 256, 4, 96), (2, 64, 4, 128), (3, 256, 4, 64), (2, 16, 4, 64), (1, 49, 4, 8)])
def test_qkv_attention(B, T, heads, ch, precision):
    from causaldiffae_amd import ops
    from oracle.unet_ref import qkv_attention
    C = heads * ch
    qkv = rnd(B, 3 * C, T, lo=-2, hi=2)                  # reference layout [B, 3C, T]
    rows = qkv.permute(0, 2, 1).contiguous().to(DEV).requires_grad_(True)
    out = ops.qkv_attention(rows, heads)                # [B, T, C]
    qc = qkv.double().requires_grad_(True)
    ref = qkv_attention(qc, heads)                      # [B, C, T]
    gy = rnd(B, C, T, seed=3)
    (out * gy.permute(0, 2, 1).to(DEV)).sum().backward()
    (ref * gy.double()).sum().backward()
    assert err(out.permute(0, 2, 1), ref) < 2e-5
    assert err(rows.grad.permute(0, 2, 1), qc.grad) < 1e-4


# ------------------------------------------------------------------ encoder BN + LeakyReLU
@pytest.mark.parametrize("training", [True, False])
def test_bn_lrelu(training):
    from causaldiffae_amd import ops
    N, C, H = 4, 32, 8
    x = rnd(N, C, H, H, lo=-1, hi=2)
    g, b = rnd(C, seed=1) + 1.5, rnd(C, seed=2)
    rm, rv = rnd(C, seed=3) * 0.1, rnd(C, seed=4) * 0.2 + 1.0
    bn = torch.nn.BatchNorm2d(C).double()
    bn.weight.data, bn.bias.data = g.double(), b.double()
    bn.running_mean.data, bn.running_var.data = rm.double().clone(), rv.double().clone()
    bn.train(training)
    xc = x.double().requires_grad_(True)
    yc = F.leaky_relu(bn(xc), 0.01)
    gd, bd = [t.to(DEV).requires_grad_(True) for t in (g, b)]
    xd = ops.to_nhwc(x.to(DEV)).requires_grad_(True)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    y = ops.bn_lrelu(xd, gd, bd, rmd, rvd, training)
    assert err(y, yc) < 2e-5
    if training:
        assert err(rmd, bn.running_mean) < 1e-6 and err(rvd, bn.running_var) < 1e-6
        gy = rnd(N, C, H, H, seed=6)
        (y * gy.to(DEV)).sum().backward()
        (yc * gy.double()).sum().backward()
        assert err(xd.grad, xc.grad) < 1e-4
        assert err(gd.grad, bn.weight.grad) < 1e-4 and err(bd.grad, bn.bias.grad) < 1e-4


# ------------------------------------------------------------------ sampler kernels vs oracle
def test_sampler_kernels():
    from causaldiffae_amd import script_util as su
    from oracle import diffusion_ref as D
    for rs in ("ddim100", ""):
        d = su.create_gaussian_diffusion(steps=1000, timestep_respacing=rs, rescale_timesteps=True)
        sch = D.Schedule(1000, "linear", rs, True)
        N = 6
        t = torch.tensor([0, 1, 5, d.num_timesteps // 2, d.num_timesteps - 2, d.num_timesteps - 1], dtype=torch.int64)
        x0, noise, x, eps = rnd(N, 4, 8, 8), rnd(N, 4, 8, 8, seed=1, lo=-2, hi=2), rnd(N, 4, 8, 8, seed=2, lo=-2, hi=2), rnd(N, 4, 8, 8, seed=3, lo=-2, hi=2)
        td = t.to(DEV)
        assert err(d.q_sample(x0.to(DEV), td, noise.to(DEV)), D.q_sample(sch, x0, t, noise)) == 0.0        # same fp32 ops, same order
        for eta in (0.0, 0.7):
            o = d._fused_update(True, x.to(DEV), eps.to(DEV), td, True, eta, noise.to(DEV) if eta else None)
            r = D.ddim_step(sch, eps, x, t, noise, eta)
            assert err(o["pred_xstart"], r["pred_xstart"]) == 0.0
            assert err(o["sample"], r["sample"]) < 2e-6
        o = d._fused_update(False, x.to(DEV), eps.to(DEV), td, True, 0.0, noise.to(DEV))
        r = D.p_sample_step(sch, eps, x, t, noise)
        assert err(o["sample"], r["sample"]) < 2e-6 and err(o["pred_xstart"], r["pred_xstart"]) == 0.0
        # timestep map: int64 indices bit exact, float rescale identical
        wm = d._wrap_model(lambda *a, **k: None)
        idx, seen = wm.map_timesteps(td)
        np.testing.assert_array_equal(idx.cpu().numpy(), np.array(d.timestep_map)[t.numpy()])
        np.testing.assert_array_equal(seen.cpu().numpy(), sch.model_t(t).numpy())


def test_timestep_embedding_and_small_ops():
    from causaldiffae_amd import nn as pnn, ops
    from oracle import unet_ref as U
    t = torch.tensor([0.0, 1.0, 10.0, 249.0, 990.0, 999.0, 123.5])
    assert err(pnn.timestep_embedding(t.to(DEV), 128), U.timestep_embedding(t, 128)) < 5e-6
    assert err(pnn.timestep_embedding(t.to(DEV), 33), U.timestep_embedding(t, 33)) < 5e-6
    x = rnd(5, 512, lo=-30, hi=30)
    assert err(ops.softplus_eps(x.to(DEV)), F.softplus(x) + 1e-8) < 1e-6
    assert err(ops.silu(x.to(DEV)), F.silu(x)) < 2e-6
    # the shared sigmoid (v_exp_f32 + v_rcp_f32, cdae_internal.h) over its whole range: relative error, large |x| (2^t overflows below -87),
    # zeros, denormal results; forward and derivative
    xe = torch.tensor([-1e4, -200.0, -104.0, -88.8, -87.5, -50.0, -20.0, -5.0, -1.0, -1e-3, -0.0, 0.0, 1e-3, 0.5, 1.0, 5.0, 20.0, 50.0, 88.0, 200.0, 1e4, 3e38])
    xg = xe.to(DEV).requires_grad_(True)
    y = ops.silu(xg)
    y.sum().backward()
    xr = xe.double().requires_grad_(True)
    yr = F.silu(xr)
    yr.sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(xg.grad).all()
    assert ((y.detach().cpu().double() - yr.detach()).abs() <= 4e-7 * yr.detach().abs() + 1e-30).all()          # (below -87 the result is x * 1.6e-38 instead of a denormal: absolute 1e-30)
    assert ((xg.grad.cpu().double() - xr.grad).abs() <= 1e-6 * xr.grad.abs() + 1e-30).all()
    xm = torch.linspace(-30, 30, 200001)
    ym = ops.silu(xm.to(DEV)).cpu().double()
    rel = ((ym - F.silu(xm.double())).abs() / F.silu(xm.double()).abs().clamp_min(1e-30)).max().item()
    assert rel < 6e-7, rel
    a, b = ops.to_nhwc(rnd(2, 128, 4, 4).to(DEV)), ops.to_nhwc(rnd(2, 384, 4, 4, seed=1).to(DEV))
    assert err(ops.cat_channels(a, b), torch.cat([a.cpu(), b.cpu()], 1)) == 0.0
    assert err(ops.to_nchw(a), a.cpu()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("shape,div,shift", [((37, 28, 28, 1), 255.0, 0.0), ((19, 96, 96, 4), 255.0, 0.0), ((11, 64, 64, 3), 127.5, -1.0)])
def test_device_feed_matches_host_feed(shape, div, shift):
    """HBM-resident pool + cdae_gather_u8 == the host feed (u8/div+shift, bit-exact), incl. labels and the epoch order."""
    from causaldiffae_amd import image_datasets as ds
    rng = np.random.RandomState(7)
    imgs = rng.randint(0, 256, size=shape).astype(np.uint8)
    imgs[0] = np.arange(np.prod(shape[1:]), dtype=np.int64).reshape(shape[1:]) % 256        # every byte value
    pool = ds.Pool(imgs, {"c": rng.rand(shape[0], 4).astype(np.float32), "y": rng.randint(0, 10, size=shape[0]).astype(np.int64)},
                   div=div, shift=shift)
    host, devf = ds.Feed(pool, 8, shuffle=True, seed=3), ds.Feed(pool, 8, shuffle=True, seed=3, device="cuda:0")
    for _ in range(5):
        xh, ch = next(host)
        xd, cd = next(devf)
        assert xd.is_cuda and xd.shape == xh.shape
        assert torch.equal(xd.cpu(), xh)
        for k in ch:
            assert torch.equal(cd[k].cpu(), ch[k])
    from causaldiffae_amd import ops
    assert ops.is_nhwc(xd) or shape[3] == 1


# ------------------------------------------------------------------ pre-split operand path (LDS-DMA kernel)
def _split_nhwc(x):
    """fp32 logical [N,C,H,W] (NHWC storage) -> ops.SplitAct through cdae_split_f16."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import check, lib, ptr, stream
    x = ops.to_nhwc(x)
    N, C, H, W = x.shape
    planes = torch.empty((2, N, H, W, C), dtype=torch.float16, device=x.device)
    check(lib.cdae_split_f16(ptr(x), ptr(planes[0]), ptr(planes[1]), x.numel(), stream()))
    return ops.SplitAct(planes[0], planes[1], (N, C, H, W))


@pytest.mark.gpu
def test_split_f16_planes():
    x = (torch.randn(4, 64, 8, 8, device="cuda:0") * 3).contiguous(memory_format=torch.channels_last)
    s = _split_nhwc(x)
    xr = x.permute(0, 2, 3, 1)
    hi = xr.half()
    assert torch.equal(s.hi, hi) and torch.equal(s.lo, (xr - hi.float()).half())
    rec = s.hi.float() + s.lo.float()
    assert (rec - xr).abs().max().item() <= 2.0 ** -21 * xr.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,S,stride,up,res,nchw", [
    (2, 128, 128, 16, 1, False, False, False),      # 64x64 tiles
    (8, 128, 256, 32, 1, False, True, False),       # 128x128 tiles, residual epilogue
    (3, 256, 384, 16, 2, False, False, False),      # stride 2, ragged M
    (2, 512, 256, 8, 1, True, False, False),        # fused nearest-2x upsample
    (2, 1024, 512, 8, 1, False, True, False),       # deep K -> split-K slabs
    (2, 128, 4, 32, 1, False, False, True),         # output head: Cout 4, NCHW result
    (1, 160, 96, 8, 1, False, False, False),        # Cin = 5 x 32, Cout not a tile multiple
    (2, 128, 128, 64, 1, False, False, False),      # window kernel at W = 64 (window 258 rows)
    (5, 64, 128, 8, 1, False, True, False),         # window kernel: 8x8 images, tiles span two images, ragged last tile
    (3, 96, 192, 16, 1, False, False, False),       # window kernel: Cin = 3 chunks, two n-tiles (one ragged)
    (64, 64, 128, 64, 1, False, True, False),       # 256-row tiles, tight 384-row window at W = 64 (>= 512 tiles), residual
    (260, 64, 256, 32, 1, False, False, False),     # 256-row tiles at W = 32, two n-tiles, ragged last tile (M % 256 != 0)
])
def test_conv3x3_presplit_is_bit_identical(N, Cin, Cout, S, stride, up, res, nchw):
    """The LDS-DMA kernel on pre-split planes runs the same products in the same order as the in-kernel split:
    identical bits, every geometry (and the same split-K choice)."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import get_precision, set_precision
    prev = get_precision()
    set_precision("f16x3")
    try:
        g = torch.Generator(device="cuda:0").manual_seed(5)
        x = ops.to_nhwc(torch.randn(N, Cin, S, S, device="cuda:0", generator=g))
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda:0", generator=g) / (9 * Cin) ** 0.5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device="cuda:0", generator=g)
        So = 2 * S if up else (S - 1) // stride + 1
        r = ops.to_nhwc(torch.randn(N, Cout, So, So, device="cuda:0", generator=g)) if res else None
        with torch.no_grad():
            ref = ops.conv3x3(x, w, b, res=r, stride=stride, up=up, out_nchw=nchw)
            got = ops.conv3x3_ps(_split_nhwc(x), w, b, res=r, stride=stride, up=up, out_nchw=nchw)
        assert got.shape == ref.shape and got.stride() == ref.stride()
        windowed = stride == 1 and not up and not nchw and N * So * So >= 96 and Cout >= 96      # window-resident kernel: K order (chunk, tap)
        if up:      # routed to the sub-pixel form (folded weights): same result up to fp32 rounding, see test_upconv_subpixel_*
            assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
        elif windowed:
            assert (got - ref).abs().max().item() < 4e-6 * max(1.0, ref.abs().max().item())
        else:
            assert torch.equal(got, ref)
        wf = w.float()
        exact = F.conv2d(F.interpolate(x.contiguous(), scale_factor=2, mode="nearest") if up else x.contiguous(), wf, b, stride=stride, padding=1)
        if res:
            exact = exact + r
        assert (got - exact).abs().max().item() < 2e-5 * max(1.0, exact.abs().max().item())
    finally:
        set_precision(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,S,Nf", [(2, 384, 16, 1152), (1, 512, 8, 1536), (4, 128, 8, 96)])
def test_linear_presplit_is_bit_identical(N, C, S, Nf):
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import get_precision, set_precision
    prev = get_precision()
    set_precision("f16x3")
    try:
        g = torch.Generator(device="cuda:0").manual_seed(6)
        x = ops.to_nhwc(torch.randn(N, C, S, S, device="cuda:0", generator=g))
        w = torch.randn(Nf, C, 1, device="cuda:0", generator=g) / C ** 0.5
        b = torch.randn(Nf, device="cuda:0", generator=g)
        rows = x.permute(0, 2, 3, 1).reshape(N * S * S, C)
        with torch.no_grad():
            ref = ops.linear(rows, w, b)
            got = ops.linear_ps(_split_nhwc(x), w, b)
        assert torch.equal(got, ref)
    finally:
        set_precision(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("C,S,ss,silu", [(128, 16, False, True), (384, 8, True, True), (512, 8, False, False)])
def test_group_norm_split_matches_fp32_output(C, S, ss, silu):
    from causaldiffae_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(7)
    x = ops.to_nhwc(torch.randn(3, C, S, S, device="cuda:0", generator=g) * 2 + 0.5)
    gamma, beta = torch.randn(C, device="cuda:0", generator=g), torch.randn(C, device="cuda:0", generator=g)
    sc = torch.randn(3, 2 * C, device="cuda:0", generator=g) * 0.3 if ss else None
    with torch.no_grad():
        y = ops.group_norm(x, gamma, beta, sc, silu).permute(0, 2, 3, 1)
        s = ops.group_norm_split(x, gamma, beta, sc, silu)
    hi = y.half()
    assert torch.equal(s.hi, hi) and torch.equal(s.lo, (y - hi.float()).half())


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,S", [(2, 256, 256, 16), (3, 512, 512, 8), (1, 384, 384, 16), (2, 128, 96, 8),
                                         (64, 256, 256, 16),      # phases on the window kernel (128-row tiles)
                                         (256, 64, 128, 32)])     # phases on the window kernel with 256-row tiles
def test_upconv_subpixel_matches_upsample_conv(N, Cin, Cout, S):
    """nearest-2x + conv3x3 as four folded 2x2 convolutions of the low-resolution input == convolving the upsampled image
    (fp32 rounding of the folded weights only)."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import get_precision, set_precision
    prev = get_precision()
    set_precision("f16x3")
    try:
        g = torch.Generator(device="cuda:0").manual_seed(8)
        x = ops.to_nhwc(torch.randn(N, Cin, S, S, device="cuda:0", generator=g))
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda:0", generator=g) / (9 * Cin) ** 0.5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device="cuda:0", generator=g)
        with torch.no_grad():
            ref = ops.conv3x3(x, w, b, up=True)
            got = ops.upconv3x3_ps(_split_nhwc(x), w, b)
            got2 = ops.conv3x3_ps(_split_nhwc(x), w, b, up=True)              # routes to the sub-pixel form
        exact = F.conv2d(F.interpolate(x.contiguous().double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
        assert got.shape == ref.shape and got.stride() == ref.stride() and torch.equal(got, got2)
        scale = max(1.0, exact.abs().max().item())
        assert (got.double() - exact).abs().max().item() < 2e-5 * scale
        assert (got - ref).abs().max().item() < 2e-5 * scale
    finally:
        set_precision(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_upconv_subpixel_random_shapes(seed, expect_kernels):
    """Seeded fuzz of the fused sub-pixel up-conv (four 2 x 2 phases of the window kernel in one launch, with and without the epilogue
    statistics), forced by the dispatch threshold, against upsample + conv2d in fp64; the GroupNorm partial sums must add up to the
    per-image sums of the result."""
    import random
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import tune_scope, range_check
    rng = random.Random(4000 + seed)
    g = torch.Generator(device="cuda:0").manual_seed(55 + seed)
    for _ in range(5):
        S = rng.choice([8, 16, 32])
        N = rng.randint(1, 6 if S == 32 else 24)
        ci, co = 32 * rng.randint(1, 10), 32 * rng.randint(1, 10)
        stats = rng.random() < 0.5
        x = ops.to_nhwc(torch.randn(N, ci, S, S, device="cuda:0", generator=g))
        w = (torch.randn(co, ci, 3, 3, device="cuda:0", generator=g) / (9 * ci) ** 0.5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device="cuda:0", generator=g)
        with torch.no_grad(), tune_scope(convwin_min_tiles=1), expect_kernels(convwin_up=1):
            y = ops.upconv3x3_ps(_split_nhwc(x), w, b, gn_stats=stats)
        exact = F.conv2d(F.interpolate(x.contiguous().double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
        e = (y.double() - exact).abs().max().item() / max(1.0, exact.abs().max().item())
        assert torch.isfinite(y).all() and e < 2e-5, ((N, ci, co, S, stats), e)
        if hasattr(y, "_gnparts"):
            parts = y._gnparts.double().reshape(4, N, -1, co, 2).sum((0, 2))
            yr = y.permute(0, 2, 3, 1).reshape(N, -1, co).double()
            assert (parts[:, :, 0] - yr.sum(1)).abs().max().item() < 1e-4 * yr.shape[1] / 32, (N, ci, co, S)
            assert (parts[:, :, 1] - (yr * yr).sum(1)).abs().max().item() < 1e-3 * yr.shape[1] / 32, (N, ci, co, S)
    range_check("upconv random shapes")


@pytest.mark.gpu
def test_epilogue_plane_outputs():
    """conv / linear epilogues that also emit the result as f16 hi/lo planes (input of a following pre-split conv)."""
    from causaldiffae_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(9)
    x = ops.to_nhwc(torch.randn(2, 128, 16, 16, device="cuda:0", generator=g))
    w = (torch.randn(256, 128, 3, 3, device="cuda:0", generator=g) / 34.0).contiguous(memory_format=torch.channels_last)
    b = torch.randn(256, device="cuda:0", generator=g)
    r = ops.to_nhwc(torch.randn(2, 256, 16, 16, device="cuda:0", generator=g))
    with torch.no_grad():
        plain = ops.conv3x3_ps(_split_nhwc(x), w, b, res=r)
        out = ops.conv3x3_ps(_split_nhwc(x), w, b, res=r, emit_split=True)
    assert torch.equal(out, plain)
    y = out.permute(0, 2, 3, 1)
    hi = y.half()
    assert torch.equal(out._split.hi, hi) and torch.equal(out._split.lo, (y - hi.float()).half())
    rows = torch.randn(2 * 8 * 8, 384, device="cuda:0", generator=g)
    wl = torch.randn(384, 384, 1, device="cuda:0", generator=g) / 20.0
    bl = torch.randn(384, device="cuda:0", generator=g)
    res = torch.randn(2 * 8 * 8, 384, device="cuda:0", generator=g)
    with torch.no_grad():
        ref = ops.linear(rows, wl, bl, res=res)
        y2, planes = ops.linear_emit(rows, wl, bl, res, (2, 384, 8, 8))
    assert torch.equal(y2, ref)
    hi = ref.half()
    assert torch.equal(planes.hi.reshape(-1, 384), hi) and torch.equal(planes.lo.reshape(-1, 384), (ref - hi.float()).half())


@pytest.mark.gpu
@pytest.mark.parametrize("C1,C2,S,ss", [(512, 384, 8, True), (256, 128, 16, False), (128, 128, 32, True)])
def test_virtual_concat_consumers(C1, C2, S, ss):
    """GroupNorm and the 1x1 skip conv reading th.cat([a, b], 1) in place (ops.CatAct) == the same ops on the materialised tensor
    (groups may straddle the seam: 896 channels / 32 groups = 28 per group)."""
    from causaldiffae_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(11)
    a = ops.to_nhwc(torch.randn(3, C1, S, S, device="cuda:0", generator=g) * 1.5 + 0.3)
    b = ops.to_nhwc(torch.randn(3, C2, S, S, device="cuda:0", generator=g) * 0.7 - 0.2)
    C = C1 + C2
    gamma, beta = torch.randn(C, device="cuda:0", generator=g), torch.randn(C, device="cuda:0", generator=g)
    sc = torch.randn(3, 2 * C, device="cuda:0", generator=g) * 0.3 if ss else None
    w = torch.randn(256, C, 1, 1, device="cuda:0", generator=g) / C ** 0.5
    bias = torch.randn(256, device="cuda:0", generator=g)
    with torch.no_grad():
        full = torch.cat([a, b], dim=1)
        ref = ops.group_norm_split(full, gamma, beta, sc, True)
        got = ops.group_norm_split(ops.CatAct(a, b), gamma, beta, sc, True)
        assert torch.equal(got.hi, ref.hi) and torch.equal(got.lo, ref.lo)
        yref = ops.conv1x1(full, w, bias)
        ygot = ops.conv1x1_cat(ops.CatAct(a, b), w, bias)
    assert ygot.shape == yref.shape
    assert (ygot - yref).abs().max().item() < 2e-6 * max(1.0, yref.abs().max().item())       # K summed in two parts


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,S,stride,cat", [(64, 128, 128, 32, 1, False), (96, 256, 384, 16, 1, True), (64, 128, 256, 32, 2, False)])
def test_groupnorm_statistics_from_conv_epilogue(N, Cin, Cout, S, stride, cat):
    """GroupNorm fed by the partial sums a conv epilogue leaves behind == GroupNorm with its own statistics pass (also for the
    channel concatenation of two such tensors, with groups straddling the seam)."""
    from causaldiffae_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(12)
    x = ops.to_nhwc(torch.randn(N, Cin, S, S, device="cuda:0", generator=g))
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda:0", generator=g) / (9 * Cin) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device="cuda:0", generator=g)
    with torch.no_grad():
        y = ops.conv3x3_ps(_split_nhwc(x), w, b, stride=stride, gn_stats=True)
        assert hasattr(y, "_gnparts")
        plain = ops.conv3x3_ps(_split_nhwc(x), w, b, stride=stride)
        # the statistics epilogue excludes split-K, so the two launches may partition K differently: fp32 rounding of the sum only
        assert (y - plain).abs().max().item() < 4e-6 * max(1.0, plain.abs().max().item())
        So = y.shape[2]
        # per-(32-row chunk, column) slots; a kernel may put the sum of a group of chunks of one image into the group's first slot and
        # zeros into the others (convwin_kernel: 128 rows, 64 for 8 x 8 images): what the consumer adds up is the sum per image
        parts = y._gnparts.double().reshape(N, -1, Cout, 2).sum(1)
        yr = y.permute(0, 2, 3, 1).reshape(N, -1, Cout).double()
        assert (parts[:, :, 0] - yr.sum(1)).abs().max().item() < 1e-4 * yr.shape[1] / 32 and (parts[:, :, 1] - (yr * yr).sum(1)).abs().max().item() < 1e-3 * yr.shape[1] / 32
        if cat:
            w2 = (torch.randn(128, Cin, 3, 3, device="cuda:0", generator=g) / (9 * Cin) ** 0.5).contiguous(memory_format=torch.channels_last)
            y2 = ops.conv3x3_ps(_split_nhwc(x), w2, None, gn_stats=True)
            src, C = ops.CatAct(y, y2), Cout + 128
            ref_in = ops.CatAct(y.detach().clone(), y2.detach().clone())        # clones carry no partial sums: statistics pass
        else:
            src, C = y, Cout
            ref_in = y.detach().clone()
        gamma, beta = torch.randn(C, device="cuda:0", generator=g), torch.randn(C, device="cuda:0", generator=g)
        got = ops.group_norm_split(src, gamma, beta, None, True)
        ref = ops.group_norm_split(ref_in, gamma, beta, None, True)
    d = ((got.hi.float() + got.lo.float()) - (ref.hi.float() + ref.lo.float())).abs().max().item()
    assert d < 2e-5, d


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_window_conv_random_shapes(seed, expect_kernels):
    """Seeded fuzz of the window conv kernel's geometry: batch, channel counts (multiples of 16 / 32 that give partial n-tiles and odd
    chunk counts), image side, residual, K split — every case forced onto convwin_kernel by the dispatch threshold and checked against
    F.conv2d in fp64 at the bar of the fixed cases (2e-5 of the largest output)."""
    import random
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import check, lib, ptr, stream, tune_scope, range_check
    rng = random.Random(1000 + seed)
    g = torch.Generator(device="cuda:0").manual_seed(77 + seed)
    for _ in range(8):
        S = rng.choice([8, 16, 32, 64])
        N = rng.randint(1, 9 if S >= 32 else 40)
        ci, co = 32 * rng.randint(1, 12), 16 * rng.randint(2, 24)
        res, splitk = rng.random() < 0.5, rng.randint(0, 1)
        x = ops.to_nhwc(torch.randn(N, ci, S, S, device="cuda:0", generator=g))
        w = (torch.randn(co, ci, 3, 3, device="cuda:0", generator=g) / (9 * ci) ** 0.5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device="cuda:0", generator=g)
        r = ops.to_nhwc(torch.randn(N, co, S, S, device="cuda:0", generator=g)) if res else None
        planes = torch.empty((2, N, S, S, ci), dtype=torch.float16, device="cuda:0")
        check(lib.cdae_split_f16(ptr(x), ptr(planes[0]), ptr(planes[1]), x.numel(), stream()))
        with torch.no_grad(), tune_scope(convwin_min_tiles=1, convwin_splitk=splitk), expect_kernels(convwin=1):
            y = ops.conv3x3_ps(ops.SplitAct(planes[0], planes[1], (N, ci, S, S)), w, b, res=r)
        exact = F.conv2d(x.double().contiguous(), w.double(), b.double(), padding=1)
        if res:
            exact = exact + r.double()
        e = (y.double() - exact).abs().max().item() / max(1.0, exact.abs().max().item())
        assert torch.isfinite(y).all() and e < 2e-5, ((N, ci, co, S, res, splitk), e)
    range_check("random shapes")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_window_conv_96_column_tiles(seed, expect_kernels):
    """convwin_kernel<f16, 9, pairs, NJ = 3>: 256 x 96 tiles (the dispatcher picks them for Cout = 384 at 16 x 16 and batch 128, where
    they fill the 512 block slots and 128-column tiles fill 384).  Forced here (cdae_tune_set CONVWIN_NJ3 = 1) onto small cases with
    Cout in {96, 192, 288, 384}: against F.conv2d in fp64, bit-identical to the 128-column instantiation (same K order per output
    element), with residual, partial row tiles, and the epilogue's GroupNorm partial sums equal to those of the 128-column kernel."""
    import random
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import tune_scope, range_check
    rng = random.Random(500 + seed)
    g = torch.Generator(device="cuda:0").manual_seed(91 + seed)
    for case in range(6):
        S = rng.choice([8, 16, 32])
        N = rng.randint(1, 6 if S >= 32 else 24)
        ci, co = 32 * rng.randint(1, 8), 96 * rng.randint(1, 4)
        res = rng.random() < 0.5
        stats = case % 2 == 0 and (S * S) % 32 == 0
        x = ops.to_nhwc(torch.randn(N, ci, S, S, device="cuda:0", generator=g))
        w = (torch.randn(co, ci, 3, 3, device="cuda:0", generator=g) / (9 * ci) ** 0.5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device="cuda:0", generator=g)
        r = ops.to_nhwc(torch.randn(N, co, S, S, device="cuda:0", generator=g)) if res else None
        xs = _split_nhwc(x)
        with torch.no_grad(), tune_scope(convwin_min_tiles=1, convwin_splitk=0):
            with tune_scope(convwin_nj3=1), expect_kernels(convwin=1):
                y3 = ops.conv3x3_ps(xs, w, b, res=r, gn_stats=stats)
            with tune_scope(convwin_nj3=-1), expect_kernels(convwin=1):
                y4 = ops.conv3x3_ps(xs, w, b, res=r, gn_stats=stats)
        exact = F.conv2d(x.double().contiguous(), w.double(), b.double(), padding=1)
        if res:
            exact = exact + r.double()
        e = (y3.double() - exact).abs().max().item() / max(1.0, exact.abs().max().item())
        assert torch.isfinite(y3).all() and e < 2e-5, ((N, ci, co, S, res), e)
        assert torch.equal(y3, y4), (N, ci, co, S, res)
        if stats and hasattr(y4, "_gnparts"):
            assert hasattr(y3, "_gnparts") and torch.equal(y3._gnparts, y4._gnparts)
    range_check("96-column tiles")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_window_conv_64_column_tiles(seed, expect_kernels):
    """convwin_kernel<f16, 9, pairs, NJ = 2>: 256 x 64 tiles (round 6: the dispatcher picks them where 256 x 128 tiles leave more than 0.4 of the
    block slots empty — small-batch sampling at 64 x 64 / 32 x 32).  Forced here (cdae_tune_set CONVWIN_NJ2 = 1) onto small cases with
    Cout in {64 .. 384}: against F.conv2d in fp64, bit-identical to the 128-column instantiation, with residual, partial row tiles, Cout not a
    multiple of 128, and the epilogue's GroupNorm partial sums equal to those of the 128-column kernel."""
    import random
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import tune_scope, range_check
    rng = random.Random(700 + seed)
    g = torch.Generator(device="cuda:0").manual_seed(191 + seed)
    for case in range(6):
        S = rng.choice([8, 16, 32])
        N = rng.randint(1, 6 if S >= 32 else 24)
        ci, co = 32 * rng.randint(1, 8), 64 * rng.randint(1, 6)
        res = rng.random() < 0.5
        stats = case % 2 == 0 and (S * S) % 32 == 0
        x = ops.to_nhwc(torch.randn(N, ci, S, S, device="cuda:0", generator=g))
        w = (torch.randn(co, ci, 3, 3, device="cuda:0", generator=g) / (9 * ci) ** 0.5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device="cuda:0", generator=g)
        r = ops.to_nhwc(torch.randn(N, co, S, S, device="cuda:0", generator=g)) if res else None
        xs = _split_nhwc(x)
        with torch.no_grad(), tune_scope(convwin_min_tiles=1, convwin_splitk=0, convwin_nj3=-1):
            with tune_scope(convwin_nj2=1), expect_kernels(convwin=1):
                y2 = ops.conv3x3_ps(xs, w, b, res=r, gn_stats=stats)
            with tune_scope(convwin_nj2=-1), expect_kernels(convwin=1):
                y4 = ops.conv3x3_ps(xs, w, b, res=r, gn_stats=stats)
        exact = F.conv2d(x.double().contiguous(), w.double(), b.double(), padding=1)
        if res:
            exact = exact + r.double()
        e = (y2.double() - exact).abs().max().item() / max(1.0, exact.abs().max().item())
        assert torch.isfinite(y2).all() and e < 2e-5, ((N, ci, co, S, res), e)
        assert torch.equal(y2, y4), (N, ci, co, S, res)
        if stats and hasattr(y4, "_gnparts"):
            assert hasattr(y2, "_gnparts") and torch.equal(y2._gnparts, y4._gnparts)
    range_check("64-column tiles")


@pytest.mark.gpu
def test_groupnorm_sums_with_a_partial_last_row_tile():
    """Epilogue GroupNorm sums when M % 256 != 0 (batch 130 at 8 x 8: 32.5 row tiles of 256; round-3 advisor finding: the last tile's
    chunks beyond M used to be written past the end of the [M / 32] buffer).  A canary row behind the buffer must stay untouched, and
    every real chunk must hold the sums of its 32 rows."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import check, lib, ptr, ptr2, stream, splitk_ws, SPLITK_BYTES, tune_scope
    g = torch.Generator(device="cuda:0").manual_seed(47)
    N, ci, co, S = 130, 64, 1024, 8
    M = N * S * S
    x = ops.to_nhwc(torch.randn(N, ci, S, S, device="cuda:0", generator=g))
    w = (torch.randn(co, ci, 3, 3, device="cuda:0", generator=g) / (9 * ci) ** 0.5).contiguous(memory_format=torch.channels_last)
    xs = _split_nhwc(x)
    w_hi, w_lo, w_sc = ops.split_weight(w)
    out = ops.new_act(N, co, S, S, "cuda:0")
    chunks = M // 32
    buf = torch.full((chunks + 8, co, 2), -7.0, device="cuda:0")          # 8 canary chunks behind the real ones
    with torch.no_grad(), tune_scope(convwin_splitk=0):
        check(lib.cdae_conv3x3_fwd_psg(ptr(xs.hi), ptr(xs.lo), S * S * ci, S * ci, ci, 0, ptr(w_hi), ptr(w_lo), None, None, ptr(w_sc), None, None, ptr(out), co,
                                       0, None, None, ptr(buf), N, S, S, ci, co, 1, 0, ptr(splitk_ws(x.device)), SPLITK_BYTES, stream()))
    torch.cuda.synchronize()
    assert (buf[chunks:] == -7.0).all(), "sums written beyond the [M / 32] buffer"
    rows = out.permute(0, 2, 3, 1).reshape(chunks, 32, co).double()
    assert (buf[:chunks, :, 0].double() - rows.sum(1)).abs().max().item() < 1e-5 * rows.sum(1).abs().max().item()
    assert (buf[:chunks, :, 1].double() - (rows * rows).sum(1)).abs().max().item() < 1e-5 * (rows * rows).sum(1).abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("res", [False, True])
def test_groupnorm_sums_from_split_k_finish(res, expect_kernels):
    """Where the window kernel splits K (the 8 x 8 level at batch 128: 128 tiles for 512 block slots) the next GroupNorm's partial sums
    come from the split-K finish kernel (splitk_reduce_gn_kernel) instead of the conv epilogue.  Against the unsplit launch of the same
    conv: identical result tensor up to the slab summation order, partial sums equal to 1e-5 of their scale, and the GroupNorm planes
    computed from them equal to those from a statistics pass over the tensor."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import tune_scope
    g = torch.Generator(device="cuda:0").manual_seed(41)
    N, ci, co, S = 128, 256, 512, 8
    x = ops.to_nhwc(torch.randn(N, ci, S, S, device="cuda:0", generator=g))
    w = (torch.randn(co, ci, 3, 3, device="cuda:0", generator=g) / (9 * ci) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device="cuda:0", generator=g)
    r = ops.to_nhwc(torch.randn(N, co, S, S, device="cuda:0", generator=g)) if res else None
    gamma, beta = torch.rand(co, device="cuda:0", generator=g) + 0.5, torch.randn(co, device="cuda:0", generator=g)
    xs = _split_nhwc(x)
    with torch.no_grad():
        with expect_kernels(convwin=1):
            y_split = ops.conv3x3_ps(xs, w, b, res=r, gn_stats=True)                 # 32 x 4 tiles -> 4 K splits + the finish
        with tune_scope(convwin_splitk=0):                                           # unsplit: 128 tiles run on the small-grid plane kernel, sums from its epilogue
            y_one = ops.conv3x3_ps(xs, w, b, res=r, gn_stats=True)
        assert hasattr(y_split, "_gnparts") and hasattr(y_one, "_gnparts")
        scale = y_one.abs().max().item()
        assert (y_split - y_one).abs().max().item() < 2e-6 * scale
        ps, po = y_split._gnparts.double(), y_one._gnparts.double()
        assert ps.shape == po.shape == (N * S * S // 32, co, 2)
        assert (ps - po).abs().max().item() < 1e-5 * po.abs().max().item()
        got = ops.group_norm_split(y_split, gamma, beta, None, True)                 # statistics from the finish kernel's sums
        ref = ops.group_norm_split(y_split.detach().clone(), gamma, beta, None, True)     # a clone carries no sums: statistics pass
    d = ((got.hi.float() + got.lo.float()) - (ref.hi.float() + ref.lo.float())).abs().max().item()
    assert d < 2e-5, d


@pytest.mark.gpu
@pytest.mark.parametrize("from_parts,cat,with_ss", [(False, False, True), (False, True, False), (True, False, True), (True, True, False)])
def test_groupnorm_coefficient_table_from_statistics_launch(from_parts, cat, with_ss):
    """The per-(image, channel) (a, b) table that the statistics kernels write on request (cdae_gn_stats2_coef /
    cdae_gn_stats_from_parts_coef: one launch less per GroupNorm whose consumer applies it while streaming the rows) is bit-identical to
    the table cdae_gn_coef computes from the same statistics — for a plain tensor, the two-source concatenation and a scale-shift slice."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import check, lib, ptr, stream
    g = torch.Generator(device="cuda:0").manual_seed(14)
    N, S = 64, 32
    x = ops.to_nhwc(torch.randn(N, 128, S, S, device="cuda:0", generator=g))
    with torch.no_grad():
        if from_parts:
            w = (torch.randn(256, 128, 3, 3, device="cuda:0", generator=g) / 34.0).contiguous(memory_format=torch.channels_last)
            a = ops.conv3x3_ps(_split_nhwc(x), w, None, gn_stats=True)
            w2 = (torch.randn(128, 128, 3, 3, device="cuda:0", generator=g) / 34.0).contiguous(memory_format=torch.channels_last)
            b = ops.conv3x3_ps(_split_nhwc(x), w2, None, gn_stats=True)
            assert hasattr(a, "_gnparts") and hasattr(b, "_gnparts")
        else:
            a = ops.to_nhwc(torch.randn(N, 256, S, S, device="cuda:0", generator=g))
            b = ops.to_nhwc(torch.randn(N, 128, S, S, device="cuda:0", generator=g))
        src = ops.CatAct(a, b) if cat else a
        C = src.shape[1]
        gamma, beta = torch.randn(C, device="cuda:0", generator=g), torch.randn(C, device="cuda:0", generator=g)
        wide = torch.randn(N, 4 * C, device="cuda:0", generator=g)
        ss = wide[:, C:3 * C] if with_ss else None                  # a column slice with a row pitch of 4 C, as the batched emb GEMM hands it over
        lz = ops.group_norm_lazy(src, gamma, beta, ss, True, want_coef=True)
        assert lz.coef is not None
        got = lz.coefficients().clone()
        ref = torch.empty_like(got)
        check(lib.cdae_gn_coef(ptr(lz.stats[0]), ptr(lz.stats[1]), ptr(gamma), ptr(beta), ptr(ss), lz.ld_ss, ptr(ref), N, C, 32, stream()))
        plain = ops.group_norm_lazy(src, gamma, beta, ss, True)
        assert plain.coef is None and torch.equal(plain.stats, lz.stats) and torch.equal(plain.coefficients(), ref)
    assert torch.equal(got, ref)
    # and it is the GroupNorm: x * a + b == (x - mean) * rstd * gamma + beta (* (1 + scale) + shift)
    xs = torch.cat([a, b], dim=1) if cat else a
    mean, rstd = lz.stats[0].double().repeat_interleave(C // 32, dim=1), lz.stats[1].double().repeat_interleave(C // 32, dim=1)
    y = (xs.double() - mean[:, :, None, None]) * rstd[:, :, None, None] * gamma.double()[None, :, None, None] + beta.double()[None, :, None, None]
    if with_ss:
        y = y * (1 + ss[:, :C].double())[:, :, None, None] + ss[:, C:].double()[:, :, None, None]
    z = xs.double() * got[:, :, 0].double()[:, :, None, None] + got[:, :, 1].double()[:, :, None, None]
    assert (y - z).abs().max().item() < 2e-5 * max(1.0, y.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,C1,S,bf16", [(16, 512, 0, 8, False), (16, 640, 384, 16, False), (32, 256, 0, 32, False), (3, 128, 0, 4, False),
                                            (16, 512, 256, 8, True), (32, 256, 0, 16, True), (256, 256, 0, 4, True)])
def test_groupnorm_statistics_one_launch_small_images(N, C, C1, S, bf16):
    """Tensors of a few MB whose groups hold at most 4096 channel vectors take the one-launch statistics kernel (gn_stats_group_kernel: one block per
    (image, group)) instead of partial + finalize: mean, rstd and the (a, b) table against an f64 statistic of the same rows, one and two
    sources (a group that straddles the concatenation boundary included: C1 = 384 with 20 channels per group), fp32 and bf16 rows."""
    from causaldiffae_amd._lib import check, lib, ptr, stream, workspace
    g = torch.Generator(device="cuda:0").manual_seed(19)
    dt = torch.bfloat16 if bf16 else torch.float32
    x = (torch.randn(N, S, S, C, device="cuda:0", generator=g) * 1.7 + 0.6).to(dt)
    a, b = (x[..., :C1].contiguous(), x[..., C1:].contiguous()) if C1 else (x, None)
    gamma, beta = torch.randn(C, device="cuda:0", generator=g), torch.randn(C, device="cuda:0", generator=g)
    ss = torch.randn(N, 2 * C, device="cuda:0", generator=g)
    mean, rstd = torch.empty(N, 32, device="cuda:0"), torch.empty(N, 32, device="cuda:0")
    coef = torch.empty(N, C, 2, device="cuda:0")
    ws = workspace(torch.device("cuda:0"), "gn_test", 64 << 20)
    ld1 = a.shape[-1]
    if bf16:
        check(lib.cdae_gn_stats16(ptr(a), ld1, ptr(b), b.shape[-1] if b is not None else 0, C1, N, S * S, C, 32, 1e-5, ptr(mean), ptr(rstd), ptr(gamma),
                                  ptr(beta), ptr(ss), 2 * C, ptr(coef), ptr(ws), stream()))
    else:
        check(lib.cdae_gn_stats2_coef(ptr(a), ld1, ptr(b), b.shape[-1] if b is not None else 4, C1, N, S * S, C, 32, 1e-5, ptr(mean), ptr(rstd),
                                      ptr(gamma), ptr(beta), ptr(ss), 2 * C, ptr(coef), ptr(ws), stream()))
    xd = x.double().reshape(N, S * S, 32, C // 32)
    m = xd.mean(dim=(1, 3))
    v = xd.var(dim=(1, 3), unbiased=False)
    r = 1.0 / torch.sqrt(v + 1e-5)
    assert (mean.double() - m).abs().max().item() < 2e-6 * max(1.0, m.abs().max().item())
    assert ((rstd.double() - r).abs() / r).max().item() < 2e-6
    A = (r.repeat_interleave(C // 32, dim=1) * gamma.double()[None])
    B = beta.double()[None] - m.repeat_interleave(C // 32, dim=1) * A
    A2, B2 = A * (1 + ss[:, :C].double()), B * (1 + ss[:, :C].double()) + ss[:, C:].double()
    assert (coef[:, :, 0].double() - A2).abs().max().item() < 1e-5 * max(1.0, A2.abs().max().item())
    assert (coef[:, :, 1].double() - B2).abs().max().item() < 1e-5 * max(1.0, B2.abs().max().item())
    if not bf16 and not C1:
        m2, r2 = torch.empty_like(mean), torch.empty_like(rstd)
        check(lib.cdae_gn_stats(ptr(x), N, S * S, C, C, 32, 1e-5, ptr(m2), ptr(r2), ptr(ws), stream()))
        assert torch.equal(m2, mean) and torch.equal(r2, rstd)


@pytest.mark.gpu
def test_groupnorm_statistics_from_upconv_phases():
    """A sub-pixel up-conv leaves four segments of partial sums (one per output parity); the next GroupNorm folds them."""
    from causaldiffae_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(13)
    x = ops.to_nhwc(torch.randn(64, 256, 16, 16, device="cuda:0", generator=g))
    w = (torch.randn(256, 256, 3, 3, device="cuda:0", generator=g) / 48.0).contiguous(memory_format=torch.channels_last)
    b = torch.randn(256, device="cuda:0", generator=g)
    skip = ops.to_nhwc(torch.randn(64, 128, 32, 32, device="cuda:0", generator=g))
    gamma, beta = torch.randn(384, device="cuda:0", generator=g), torch.randn(384, device="cuda:0", generator=g)
    with torch.no_grad():
        y = ops.upconv3x3_ps(_split_nhwc(x), w, b, gn_stats=True)
        plain = ops.upconv3x3_ps(_split_nhwc(x), w, b)           # may split K (the statistics epilogue cannot): fp32 rounding of the sum only
        assert y._gnseg == 4 and (y - plain).abs().max().item() < 4e-6 * max(1.0, plain.abs().max().item())
        wk = (torch.randn(128, 128, 3, 3, device="cuda:0", generator=g) / 34.0).contiguous(memory_format=torch.channels_last)
        s2 = ops.conv3x3_ps(_split_nhwc(skip), wk, None, gn_stats=True)
        got = ops.group_norm_split(ops.CatAct(y, s2), gamma, beta, None, True)
        ref = ops.group_norm_split(ops.CatAct(y.detach().clone(), s2.detach().clone()), gamma, beta, None, True)
    d = ((got.hi.float() + got.lo.float()) - (ref.hi.float() + ref.lo.float())).abs().max().item()
    assert d < 2e-5, d


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,heads,ch", [(3, 256, 4, 96), (5, 64, 4, 128), (2, 256, 4, 64), (2, 64, 2, 96), (1, 256, 1, 128)])
def test_fused_attention_matches_three_kernel_path(B, T, heads, ch):
    """QKVAttention as one kernel (probabilities in registers) vs the GEMM / softmax / GEMM path and vs an fp64 restatement."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import check, lib, ptr, stream
    g = torch.Generator(device="cuda:0").manual_seed(15)
    qkv = torch.randn(B, T, 3 * heads * ch, device="cuda:0", generator=g) * 1.2
    with torch.no_grad():
        got = ops.qkv_attention(qkv, heads)
        out = torch.empty_like(got)
        probs = torch.empty((B * heads, T, T), dtype=torch.float32, device="cuda:0")
        check(lib.cdae_qkv_attention_fwd(ptr(qkv), ptr(out), ptr(probs), B, T, heads, ch, stream()))
    x = qkv.double().reshape(B, T, heads, 3, ch)
    q, k, v = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]
    w = torch.softmax(torch.einsum("bthc,bshc->bhts", q, k) / ch ** 0.5, dim=-1)
    exact = torch.einsum("bhts,bshc->bthc", w, v).reshape(B, T, heads * ch)
    scale = exact.abs().max().item()
    assert (got.double() - exact).abs().max().item() < 2e-5 * max(1.0, scale)
    assert (got - out).abs().max().item() < 2e-5 * max(1.0, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,heads,ch,seed", [(16, 256, 4, 96, 6), (16, 256, 4, 128, 7), (64, 64, 4, 128, 8), (16, 256, 4, 64, 9)])
def test_fused_attention_rows_sum_to_one(B, T, heads, ch, seed):
    """Size-independent property: with V == 1 every output is the sum of one query's probabilities, so it must be 1 to fp32
    rounding for EVERY query.  A single inconsistent hi / lo pair in the in-register split of P shows up as 2^-16 here (this
    caught a double-rounding fold in the f16 conversion that hit about one query in a thousand)."""
    from causaldiffae_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    qkv = torch.randn(B, T, 3 * heads * ch, device="cuda:0", generator=g) * 1.2
    qkv.view(B, T, heads, 3, ch)[:, :, :, 2] = 1.0
    with torch.no_grad():
        o = ops.qkv_attention(qkv, heads)
    assert (o - 1.0).abs().max().item() < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,heads,ch", [(3, 256, 4, 96), (5, 64, 4, 128), (2, 256, 2, 64), (1, 64, 1, 96), (2, 256, 1, 128)])
def test_fused_attention_training_forward_keeps_probabilities(B, T, heads, ch):
    """The training forward on the one-kernel attention: the probabilities it writes for the backward, its output and the gradient of
    qkv through cdae_qkv_attention_bwd, against an fp64 restatement of reference unet.py:239-253."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import check, lib, ptr, stream
    g = torch.Generator(device="cuda:0").manual_seed(21)
    qkv = (torch.randn(B, T, 3 * heads * ch, device="cuda:0", generator=g) * 1.1).requires_grad_(True)
    assert ops._FUSED_ATTN_TRAIN and lib.cdae_qkv_attention_fused_supported(T, ch)
    out = ops.qkv_attention(qkv, heads)
    go = torch.randn(out.shape, device="cuda:0", generator=g)
    out.backward(go)
    probs = torch.empty((B * heads, T, T), dtype=torch.float32, device="cuda:0")
    o2 = torch.empty_like(out)
    check(lib.cdae_qkv_attention_fwd_fused_p(ptr(qkv.detach()), ptr(o2), ptr(probs), B, T, heads, ch, stream()))
    xd = qkv.detach().double().requires_grad_(True)
    x = xd.reshape(B, T, heads, 3, ch)
    q, k, v = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]
    w = torch.softmax(torch.einsum("bthc,bshc->bhts", q, k) / ch ** 0.5, dim=-1)
    exact = torch.einsum("bhts,bshc->bthc", w, v).reshape(B, T, heads * ch)
    exact.backward(go.double())
    assert torch.equal(o2, out.detach())
    assert (probs.double() - w.detach().reshape(B * heads, T, T)).abs().max().item() < 2e-6
    assert (probs.sum(-1) - 1.0).abs().max().item() < 2e-6
    assert (out.detach().double() - exact.detach()).abs().max().item() < 2e-5 * max(1.0, exact.abs().max().item())
    assert (qkv.grad.double() - xd.grad).abs().max().item() < 2e-4 * xd.grad.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,heads,ch", [(3, 256, 4, 96), (5, 64, 4, 128), (2, 256, 2, 64), (2, 64, 1, 96), (1, 256, 1, 128), (2, 64, 2, 64)])
def test_attention_backward_query_side_in_one_kernel(B, T, heads, ch):
    """cdae_qkv_attention_bwd_q_fused (dP = dO V^T, softmax backward, dQ = dS K / sqrt(ch) in one launch) against fp64: dS and the q slices
    of dqkv, every (T, head dim) instantiation — the dispatcher itself only takes it for T = 64."""
    from causaldiffae_amd._lib import check, lib, ptr, stream
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(31)
    qkv = torch.randn(B, T, 3 * heads * ch, device=dev, generator=g) * 1.1
    dout = torch.randn(B, T, heads * ch, device=dev, generator=g) * 1e-3            # gradient-sized values (bf16 planes: no underflow)
    x = qkv.double().reshape(B, T, heads, 3, ch)
    q, k, v = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]
    w = torch.softmax(torch.einsum("bthc,bshc->bhts", q, k) / ch ** 0.5, dim=-1)
    go = dout.double().reshape(B, T, heads, ch)
    dp = torch.einsum("bthc,bshc->bhts", go, v)
    ds = w * (dp - (w * dp).sum(-1, keepdim=True))
    dq = torch.einsum("bhts,bshc->bthc", ds, k) / ch ** 0.5
    probs = w.float().reshape(B * heads, T, T).contiguous()
    dqkv = torch.zeros_like(qkv)
    ds_out = torch.empty_like(probs)
    check(lib.cdae_qkv_attention_bwd_q_fused(ptr(qkv), ptr(probs), ptr(dout), ptr(dqkv), ptr(ds_out), B, T, heads, ch, stream()))
    got_dq = dqkv.reshape(B, T, heads, 3, ch)[:, :, :, 0].double()
    assert (ds_out.double() - ds.reshape(B * heads, T, T)).abs().max().item() < 1e-4 * ds.abs().max().item()
    assert (got_dq - dq).abs().max().item() < 1e-4 * dq.abs().max().item()
    assert dqkv.reshape(B, T, heads, 3, ch)[:, :, :, 1:].abs().max().item() == 0.0       # k / v slices untouched


# ----------------------------------------------------------------------------- training on the pre-split kernels
def _wgrad_ref(a, dy):
    """fp64 weight / bias gradient of a stride-1 conv3x3 (NCHW a [N,Cin,H,W], dy [N,Cout,H,W]) in OHWI order."""
    N, Cin, H, W = a.shape
    Cout = dy.shape[1]
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, device=a.device, requires_grad=True)
    out = torch.nn.functional.conv2d(a.double(), w, None, padding=1)
    out.backward(dy.double())
    return w.grad.permute(0, 2, 3, 1).contiguous(), dy.double().sum(dim=(0, 2, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,Cin,Cout,accumulate", [
    (3, 8, 8, 64, 64, 0),          # one 64-pixel step per image: every step crosses an image boundary; one block per tile (no split-K)
    (5, 8, 8, 128, 64, 1),         # odd image count, accumulate into an existing gradient
    (2, 16, 16, 64, 128, 0),       # 4 steps per image
    (3, 32, 32, 128, 128, 0),      # split-K over pixel ranges that start inside an image
    (2, 64, 64, 64, 64, 1),        # one image row per step, widest ring
    (4, 16, 8, 64, 64, 0),         # non-square image (H != W)
    (2, 16, 16, 640, 384, 0),      # skip-concatenation widths: 10 x 6 channel tiles
    (1, 8, 8, 1024, 512, 1),       # widest layer of the network, a single image
    (12, 16, 16, 256, 256, 0),     # 3 steps per block against 4 steps per image: blocks start and end at every phase inside an image
    (10, 32, 32, 256, 128, 0),     # 5 steps per block against 16 per image
    (37, 8, 8, 64, 64, 0),         # one step per block, a prime number of images
])
def test_wgrad_window_kernel_matches_fp64(N, H, W, Cin, Cout, accumulate):
    """cdae_conv3x3_wgrad_win (LDS-ring window, transpose-read fragments, bf16 hi/lo planes) against autograd in fp64; operands are
    exactly representable as hi + lo so the comparison sees only the fp32 accumulation."""
    _wgrad_window_case(N, H, W, Cin, Cout, accumulate)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_window_dgrad_random_shapes(seed, expect_kernels):
    """Seeded fuzz of the data-gradient instantiation of the window conv kernel (convwin_kernel<bf16>: bf16 hi/lo planes of dy, flipped-tap
    [Cin][9][Cout] weight planes in K-group-major order), forced by the dispatch threshold with and without its K split, against
    torch.nn.grad.conv2d_input in fp64.  dy is exactly hi + lo, the weight is rounded to bf16 x 2 (2^-16): bar 3e-5 of the largest value."""
    import random
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import check, lib, ptr, ptr2, stream, tune_scope, splitk_ws, SPLITK_BYTES, range_check
    rng = random.Random(3000 + seed)
    g = torch.Generator(device="cuda:0").manual_seed(91 + seed)
    dev = torch.device("cuda:0")
    for _ in range(6):
        S = rng.choice([8, 16, 32, 64])
        N = rng.randint(1, 6 if S >= 32 else 36)
        ci, co = 16 * rng.randint(2, 20), 32 * rng.randint(1, 10)
        splitk = rng.randint(0, 1)
        w = (torch.randn(co, ci, 3, 3, device=dev, generator=g) / (9 * co) ** 0.5).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, S, S, co, device=dev, generator=g) * 1e-3
        dp = torch.empty((2, N, S, S, co), dtype=torch.bfloat16, device=dev)
        check(lib.cdae_split_bf16(ptr(dy), ptr(dp[0]), ptr(dp[1]), dy.numel(), stream()))
        dy_q = (dp[0].double() + dp[1].double()).permute(0, 3, 1, 2).contiguous()
        dx = ops.new_act(N, ci, S, S, dev)
        with tune_scope(convwin_min_tiles=1, convwin_splitk=splitk), expect_kernels(convwin_dgrad=1):
            check(lib.cdae_conv3x3_dgrad_psk(*ptr2(dp), *ops._wptrs(w, True), ptr(dx), ci, N, S, S, ci, co, ptr(splitk_ws(dev)), SPLITK_BYTES, stream()))
        exact = torch.nn.grad.conv2d_input((N, ci, S, S), w.double().contiguous(), dy_q, padding=1)
        e = (dx.double() - exact).abs().max().item() / exact.abs().max().item()
        assert torch.isfinite(dx).all() and e < 3e-5, ((N, ci, co, S, splitk), e)
    range_check("dgrad random shapes")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_wgrad_window_kernel_random_shapes(seed):
    """Seeded fuzz of the window wgrad kernel's geometry (image count, side, non-square images, channel tiles, accumulate): the ring,
    the image gaps, the mirror slots and the K split see block boundaries at every phase."""
    import random
    rng = random.Random(2000 + seed)
    for _ in range(8):
        W = rng.choice([8, 16, 32, 64])
        H = rng.choice([h for h in (8, 16, 32, 64) if (h * W) % 64 == 0 and h * W <= 4096])
        N = rng.randint(1, 6 if H * W >= 1024 else 48)
        _wgrad_window_case(N, H, W, 64 * rng.randint(1, 6), 64 * rng.randint(1, 6), rng.randint(0, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_wgrad_window_group_launch(seed):
    """cdae_conv3x3_wgrad_win_group: several weight gradients of DIFFERENT shapes (channel tiles, image sides, batch) in one launch —
    descriptors in the kernel arguments, every block finds its own; the group's splits are sized together, so some members run split
    (slabs + finish) and some unsplit in the same launch; more than twelve items go out as several launches.  Each member against
    autograd in fp64 at the single launch's bar, accumulate on and off."""
    import ctypes, random
    from causaldiffae_amd._lib import WgItem, check, lib, ptr, stream, splitk_ws, SPLITK_BYTES
    dev = "cuda:0"
    rng = random.Random(500 + seed)
    g = torch.Generator(device=dev).manual_seed(40 + seed)
    n_items = [3, 7, 14][seed]
    items, keep, checks = [], [], []
    for i in range(n_items):
        W = rng.choice([8, 16, 32])
        H = W
        N = rng.randint(1, 3 if W == 32 else 12)
        Cin, Cout, acc = 64 * rng.randint(1, 4), 64 * rng.randint(1, 4), rng.randint(0, 1)
        a = torch.randn(N, H, W, Cin, device=dev, generator=g)
        dy = torch.randn(N, H, W, Cout, device=dev, generator=g) * 1e-3
        ap = torch.empty((2, N, H, W, Cin), dtype=torch.bfloat16, device=dev)
        dp = torch.empty((2, N, H, W, Cout), dtype=torch.bfloat16, device=dev)
        check(lib.cdae_split_bf16(ptr(a), ptr(ap[0]), ptr(ap[1]), a.numel(), stream()))
        check(lib.cdae_split_bf16(ptr(dy), ptr(dp[0]), ptr(dp[1]), dy.numel(), stream()))
        dw0 = torch.randn(Cout, 3, 3, Cin, device=dev, generator=g) * 1e-2
        db0 = torch.randn(Cout, device=dev, generator=g) * 1e-2
        dw, db = dw0.clone(), db0.clone()
        items.append(WgItem(ptr(ap[0]), ptr(ap[1]), ptr(dp[0]), ptr(dp[1]), ptr(dw), ptr(db), N, H, W, Cin, Cout, acc))
        keep.append((ap, dp))
        checks.append((ap, dp, dw0, db0, dw, db, acc))
    arr = (WgItem * n_items)(*items)
    check(lib.cdae_conv3x3_wgrad_win_group(arr, n_items, ptr(splitk_ws(torch.device(dev))), SPLITK_BYTES, stream()))
    torch.cuda.synchronize()
    for ap, dp, dw0, db0, dw, db, acc in checks:
        a_q = (ap[0].float() + ap[1].float()).permute(0, 3, 1, 2)
        dy_q = (dp[0].float() + dp[1].float()).permute(0, 3, 1, 2)
        ref_w, ref_b = _wgrad_ref(a_q, dy_q)
        if acc:
            ref_w, ref_b = ref_w + dw0.double(), ref_b + db0.double()
        assert (dw.double() - ref_w).abs().max().item() < 3e-5 * ref_w.abs().max().item()
        assert (db.double() - ref_b).abs().max().item() < 3e-5 * ref_b.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f16x3", "mixed16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(5, 32, 32, 128, 64), (3, 64, 64, 64, 128), (9, 16, 16, 192, 128), (21, 8, 8, 128, 128), (4, 16, 32, 64, 64),
                                            (2, 32, 16, 64, 64)])
def test_wgrad_window_schedule_and_layout_are_bit_identical(N, H, W, Cin, Cout, mode):
    """Round 6: the window weight gradient requests its operands two steps ahead (deeper ring, three dy stages, counted waits) and keeps the
    32-byte halves of LDS rows with bit 3 set swapped (DMA-time permutation, conflict-free transpose reads).  Neither may change a bit:
    every (distance, layout) combination against the round-5 form (distance 1, plain rows), on two operand planes and on one (`mixed16`),
    shapes with image gaps inside a block's range, W = 8 (no swap), W = 64 (distance 1 on two planes: the LDS does not hold the ring),
    non-square images both ways; and the default against fp64."""
    from causaldiffae_amd._lib import check, lib, ptr, stream, splitk_ws, SPLITK_BYTES, precision_scope, tune_scope
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(77)
    a = torch.randn(N, H, W, Cin, device=dev, generator=g)
    dy = torch.randn(N, H, W, Cout, device=dev, generator=g) * 1e-3
    ap = torch.empty((2, N, H, W, Cin), dtype=torch.bfloat16, device=dev)
    dp = torch.empty((2, N, H, W, Cout), dtype=torch.bfloat16, device=dev)
    check(lib.cdae_split_bf16(ptr(a), ptr(ap[0]), ptr(ap[1]), a.numel(), stream()))
    check(lib.cdae_split_bf16(ptr(dy), ptr(dp[0]), ptr(dp[1]), dy.numel(), stream()))
    out = {}
    with precision_scope(mode):
        for dist, swz in ((1, 0), (2, 1), (2, 0), (1, 1)):
            dw = torch.full((Cout, 3, 3, Cin), float("nan"), device=dev)
            db = torch.full((Cout,), float("nan"), device=dev)
            with tune_scope(wgwin_dist=dist, wgwin_swz=swz):
                check(lib.cdae_conv3x3_wgrad_win(ptr(ap[0]), ptr(ap[1]), ptr(dp[0]), ptr(dp[1]), ptr(dw), ptr(db), N, H, W, Cin, Cout, 0,
                                                 ptr(splitk_ws(torch.device(dev))), SPLITK_BYTES, stream()))
            torch.cuda.synchronize()
            out[(dist, swz)] = (dw, db)
    base = out[(1, 0)]
    for k, (dw, db) in out.items():
        assert torch.equal(dw, base[0]), k
        assert (db - base[1]).abs().max().item() <= 1e-5 * base[1].abs().max().item(), k      # (column sums arrive by atomicAdd: order of the blocks)
    planes = 1 if mode == "mixed16" else 2
    a_q = sum(ap[i].float() for i in range(planes)).permute(0, 3, 1, 2)
    dy_q = sum(dp[i].float() for i in range(planes)).permute(0, 3, 1, 2)
    ref_w, _ = _wgrad_ref(a_q, dy_q)
    assert (out[(2, 1)][0].double() - ref_w).abs().max().item() < 3e-5 * ref_w.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,Cin,Cout,acc", [(5, 32, 32, 128, 128, 0), (3, 64, 64, 64, 256, 1), (9, 16, 16, 192, 384, 0), (21, 8, 8, 128, 128, 1), (4, 16, 32, 64, 128, 0),
                                                (37, 8, 8, 64, 256, 0), (12, 16, 16, 256, 256, 0)])
def test_wgrad_window_128_channel_block_tiles(N, H, W, Cin, Cout, acc):
    """wgwin_kernel<1 plane, CO2> (tune key wgwin_co2): the block owns 128 output channels, its two wave groups split them instead of the step's
    pixels (every wave walks both pixel halves; no fold through LDS; dy in four sub-planes per stage).  One bf16 plane per operand: against
    autograd in fp64 on exactly those planes at the bar of the shipped kernel, against the shipped kernel itself (a different summation order:
    1e-5), bias gradients included, with and without accumulation, split ranges that start inside images, W = 8 and W = 64."""
    from causaldiffae_amd._lib import check, lib, ptr, stream, splitk_ws, SPLITK_BYTES, precision_scope, tune_scope
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(123)
    a = torch.randn(N, H, W, Cin, device=dev, generator=g)
    dy = torch.randn(N, H, W, Cout, device=dev, generator=g) * 1e-3
    ap = torch.empty((2, N, H, W, Cin), dtype=torch.bfloat16, device=dev)
    dp = torch.empty((2, N, H, W, Cout), dtype=torch.bfloat16, device=dev)
    check(lib.cdae_split_bf16(ptr(a), ptr(ap[0]), ptr(ap[1]), a.numel(), stream()))
    check(lib.cdae_split_bf16(ptr(dy), ptr(dp[0]), ptr(dp[1]), dy.numel(), stream()))
    dw0 = torch.randn(Cout, 3, 3, Cin, device=dev, generator=g) * 1e-2
    db0 = torch.randn(Cout, device=dev, generator=g) * 1e-2
    out = {}
    with precision_scope("mixed16"):
        for co2 in (0, 1):
            dw, db = dw0.clone(), db0.clone()
            with tune_scope(wgwin_co2=2 * co2):
                check(lib.cdae_conv3x3_wgrad_win(ptr(ap[0]), ptr(ap[1]), ptr(dp[0]), ptr(dp[1]), ptr(dw), ptr(db), N, H, W, Cin, Cout, acc,
                                                 ptr(splitk_ws(torch.device(dev))), SPLITK_BYTES, stream()))
            torch.cuda.synchronize()
            out[co2] = (dw, db)
    ref_w, ref_b = _wgrad_ref(ap[0].float().permute(0, 3, 1, 2), dp[0].float().permute(0, 3, 1, 2))
    if acc:
        ref_w, ref_b = ref_w + dw0.double(), ref_b + db0.double()
    for co2 in (0, 1):
        assert (out[co2][0].double() - ref_w).abs().max().item() < 3e-5 * ref_w.abs().max().item(), co2
        assert (out[co2][1].double() - ref_b).abs().max().item() < 3e-5 * ref_b.abs().max().item(), co2
    assert (out[1][0] - out[0][0]).abs().max().item() < 1e-5 * out[0][0].abs().max().item()


def _wgrad_window_case(N, H, W, Cin, Cout, accumulate):
    from causaldiffae_amd._lib import check, lib, ptr, stream, splitk_ws, SPLITK_BYTES
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(11)
    a = torch.randn(N, H, W, Cin, device=dev, generator=g)
    dy = torch.randn(N, H, W, Cout, device=dev, generator=g) * 1e-3
    ap = torch.empty((2, N, H, W, Cin), dtype=torch.bfloat16, device=dev)
    dp = torch.empty((2, N, H, W, Cout), dtype=torch.bfloat16, device=dev)
    check(lib.cdae_split_bf16(ptr(a), ptr(ap[0]), ptr(ap[1]), a.numel(), stream()))
    check(lib.cdae_split_bf16(ptr(dy), ptr(dp[0]), ptr(dp[1]), dy.numel(), stream()))
    a_q = (ap[0].float() + ap[1].float()).permute(0, 3, 1, 2)           # what the kernel sees
    dy_q = (dp[0].float() + dp[1].float()).permute(0, 3, 1, 2)
    assert (a_q.permute(0, 2, 3, 1) - a).abs().max().item() < 2e-4 * a.abs().max().item()      # bf16 hi + lo: 2^-16
    dw0 = torch.randn(Cout, 3, 3, Cin, device=dev, generator=g) * 1e-2
    db0 = torch.randn(Cout, device=dev, generator=g) * 1e-2
    dw, db = dw0.clone(), db0.clone()
    assert lib.cdae_conv3x3_wgrad_win_supported(N, H, W, Cin, Cout) == 1
    check(lib.cdae_conv3x3_wgrad_win(ptr(ap[0]), ptr(ap[1]), ptr(dp[0]), ptr(dp[1]), ptr(dw), ptr(db), N, H, W, Cin, Cout, accumulate,
                                     ptr(splitk_ws(torch.device(dev))), SPLITK_BYTES, stream()))
    ref_w, ref_b = _wgrad_ref(a_q, dy_q)
    if accumulate:
        ref_w, ref_b = ref_w + dw0.double(), ref_b + db0.double()
    # hi*hi + hi*lo + lo*hi drops lo*lo (2^-16 relative to a product) and accumulates in fp32
    assert (dw.double() - ref_w).abs().max().item() < 3e-5 * ref_w.abs().max().item()
    assert (db.double() - ref_b).abs().max().item() < 3e-5 * ref_b.abs().max().item()


@pytest.mark.gpu
def test_wgrad_window_kernel_rejects_unsupported_shapes():
    from causaldiffae_amd._lib import lib
    assert lib.cdae_conv3x3_wgrad_win_supported(2, 8, 8, 96, 64) == 0          # Cin % 64
    assert lib.cdae_conv3x3_wgrad_win_supported(2, 12, 12, 64, 64) == 0        # rows not a power of two
    assert lib.cdae_conv3x3_wgrad_win_supported(2, 4, 4, 64, 64) == 0          # fewer than 64 pixels per image
    assert lib.cdae_conv3x3_wgrad_win_supported(2, 128, 128, 64, 64) == 0      # rows wider than the ring


def _gnconv_case(N, C, Cout, H, ss_on, res_on, mode):
    from causaldiffae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    cl = torch.channels_last
    x = torch.randn(N, C, H, H, device=dev, generator=g).contiguous(memory_format=cl)
    gamma = 1 + 0.1 * torch.randn(C, device=dev, generator=g)
    beta = 0.1 * torch.randn(C, device=dev, generator=g)
    ss = 0.2 * torch.randn(N, 2 * C, device=dev, generator=g) if ss_on else None
    w = (torch.randn(Cout, C, 3, 3, device=dev, generator=g) / (3 * C ** 0.5)).contiguous(memory_format=cl)
    b = 0.1 * torch.randn(Cout, device=dev, generator=g)
    res = torch.randn(N, Cout, H, H, device=dev, generator=g).contiguous(memory_format=cl) if res_on else None
    dy = torch.randn(N, Cout, H, H, device=dev, generator=g).contiguous(memory_format=cl) * 1e-3
    dt = torch.float64 if mode == "f64" else torch.float32
    leaves = [None if t is None else t.detach().to(dt).requires_grad_() for t in (x, gamma, beta, ss, w, b, res)]
    x, gamma, beta, ss, w, b, res = leaves
    if mode == "f64":
        h = torch.nn.functional.group_norm(x, 32, gamma, beta, 1e-5)
        if ss_on:
            h = h * (1 + ss[:, :C, None, None]) + ss[:, C:, None, None]
        out = torch.nn.functional.conv2d(torch.nn.functional.silu(h), w, b, padding=1)
        out = out + res if res_on else out
    elif mode == "fused":
        out = ops.gn_conv3x3(x, gamma, beta, ss, w, b, res, True, 32, 1e-5)
    else:
        out = ops.conv3x3(ops.group_norm(x, gamma, beta, ss, True, 32, 1e-5), w, b, res)
    out.backward(dy.to(dt))
    return [out.detach()] + [t.grad for t in leaves if t is not None]


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,Cout,H,ss_on,res_on", [(4, 128, 128, 64, True, True), (8, 256, 256, 32, False, False), (8, 384, 384, 16, True, True),
                                                     (16, 512, 512, 8, True, False), (2, 128, 256, 32, True, True), (3, 896, 384, 16, True, True),
                                                     (1, 1024, 512, 8, False, True), (3, 128, 128, 8, True, True)])      # last: 192 rows, 1.5 row tiles
def test_fused_gn_conv_training_node(N, C, Cout, H, ss_on, res_on):
    """ops.gn_conv3x3 (GroupNorm -> SiLU -> conv3x3 as one autograd node on the pre-split kernels: f16-plane forward, bf16-plane
    dgrad on the window kernel, LDS-ring wgrad) against torch autograd in fp64, and no worse than the separate-node path."""
    from causaldiffae_amd import ops
    with torch.enable_grad():
        assert ops.train_presplit_ok((N, C, H, H), Cout)
        fused, old, ref = (_gnconv_case(N, C, Cout, H, ss_on, res_on, m) for m in ("fused", "old", "f64"))
    for f, o, r in zip(fused, old, ref):
        sc = r.abs().max().item() + 1e-30
        ef, eo = (f.double() - r).abs().max().item() / sc, (o.double() - r).abs().max().item() / sc
        assert ef < 2e-5, (ef, eo)                     # forward 2^-22 products, backward bf16x3 (2^-16 per product, averaged down)
        assert ef < 3 * eo + 5e-6, (ef, eo)


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H", [(4, 256, 32), (8, 512, 8), (2, 384, 16)])
def test_upsample_conv_training_node(N, C, H):
    """ops.upconv3x3_train (nearest-2x written once as operand planes, conv / dgrad / wgrad on the window kernels, 2x2 sum-pool of
    the input gradient) against interpolate + conv2d autograd in fp64."""
    from causaldiffae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(7)
    cl = torch.channels_last
    x0 = torch.randn(N, C, H, H, device=dev, generator=g).contiguous(memory_format=cl)
    w0 = (torch.randn(C, C, 3, 3, device=dev, generator=g) / (3 * C ** 0.5)).contiguous(memory_format=cl)
    b0 = 0.1 * torch.randn(C, device=dev, generator=g)
    dy = torch.randn(N, C, 2 * H, 2 * H, device=dev, generator=g).contiguous(memory_format=cl) * 1e-3
    res = []
    with torch.enable_grad():
        for mode in ("f64", "fused"):
            dt = torch.float64 if mode == "f64" else torch.float32
            x, w, b = [t.detach().to(dt).requires_grad_() for t in (x0, w0, b0)]
            if mode == "f64":
                out = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
            else:
                assert ops.upconv3x3_train_ok(x, C)
                out = ops.upconv3x3_train(x, w, b)
            out.backward(dy.to(dt))
            res.append([out.detach(), x.grad, w.grad, b.grad])
    for r, f in zip(*res):
        assert (f.double() - r).abs().max().item() < 2e-5 * r.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,Cout,H", [(4, 128, 128, 32), (2, 256, 128, 16), (8, 512, 512, 8), (2, 128, 256, 64)])
def test_resblock_training_node(N, C, Cout, H):
    """ops.resblock_train: the whole ResBlock (GN-SiLU-conv, GN-scale/shift-SiLU-conv, identity or 1x1 skip) as one autograd node —
    residual gradient folded into GroupNorm's dx kernel, the inter-conv gradient passed as bf16 planes — against torch autograd in
    fp64 for every input and parameter."""
    import torch.nn.functional as F
    from causaldiffae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(9)
    cl = torch.channels_last

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, device=dev, generator=g) * scale

    x0 = rnd(N, C, H, H).contiguous(memory_format=cl)
    ss0 = rnd(N, 2 * Cout, scale=0.2)
    p0 = dict(g1=1 + rnd(C, scale=0.1), b1=rnd(C, scale=0.1), w1=(rnd(Cout, C, 3, 3) / (3 * C ** 0.5)).contiguous(memory_format=cl),
              c1b=rnd(Cout, scale=0.1), g2=1 + rnd(Cout, scale=0.1), b2=rnd(Cout, scale=0.1),
              w2=(rnd(Cout, Cout, 3, 3) / (3 * Cout ** 0.5)).contiguous(memory_format=cl), c2b=rnd(Cout, scale=0.1))
    if C != Cout:
        p0.update(sw=rnd(Cout, C, 1, 1) / C ** 0.5, sb=rnd(Cout, scale=0.1))
    dy = rnd(N, Cout, H, H).contiguous(memory_format=cl) * 1e-3
    res = []
    with torch.enable_grad():
        for mode in ("f64", "node"):
            dt = torch.float64 if mode == "f64" else torch.float32
            x, ss = x0.detach().to(dt).requires_grad_(), ss0.detach().to(dt).requires_grad_()
            p = {k: v.detach().to(dt).requires_grad_() for k, v in p0.items()}
            if mode == "f64":
                h = F.conv2d(F.silu(F.group_norm(x, 32, p["g1"], p["b1"], 1e-5)), p["w1"], p["c1b"], padding=1)
                h = F.group_norm(h, 32, p["g2"], p["b2"], 1e-5) * (1 + ss[:, :Cout, None, None]) + ss[:, Cout:, None, None]
                out = F.conv2d(F.silu(h), p["w2"], p["c2b"], padding=1) + (x if C == Cout else F.conv2d(x, p["sw"], p["sb"]))
            else:
                out = ops.resblock_train(x, ss, p["g1"], p["b1"], p["w1"], p["c1b"], p["g2"], p["b2"], p["w2"], p["c2b"], p.get("sw"), p.get("sb"))
            out.backward(dy.to(dt))
            res.append([out.detach(), x.grad, ss.grad] + [p[k].grad for k in sorted(p)])
    names = ["out", "dx", "dss"] + sorted(p0)
    for n, r, f in zip(names, *res):
        assert (f.double().reshape(r.shape) - r).abs().max().item() < 3e-5 * r.abs().max().item(), n


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,Cin,Cout,accumulate", [(4, 16, 16, 128, 6, 0), (3, 8, 12, 64, 3, 1), (2, 32, 32, 256, 8, 0), (5, 7, 9, 36, 1, 0)])
def test_wgrad_few_output_channels(N, H, W, Cin, Cout, accumulate):
    """cdae_conv3x3_wgrad_fewout (sliding-window fp32 FMAs, the `out` conv's wgrad) against autograd in fp64, incl. odd image sizes,
    channel counts that are not a multiple of the block, and accumulation into an existing gradient."""
    from causaldiffae_amd._lib import check, lib, ptr, stream, splitk_ws, SPLITK_BYTES
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(13)
    a = torch.randn(N, H, W, Cin, device=dev, generator=g)
    dy = torch.randn(N, H, W, Cout, device=dev, generator=g) * 1e-2
    dw0 = torch.randn(Cout, 3, 3, Cin, device=dev, generator=g) * 1e-2
    db0 = torch.randn(Cout, device=dev, generator=g) * 1e-2
    dw, db = dw0.clone(), db0.clone()
    check(lib.cdae_conv3x3_wgrad_fewout(ptr(a), ptr(dy), Cout, ptr(dw), ptr(db), N, H, W, Cin, Cout, accumulate, ptr(splitk_ws(torch.device(dev))),
                                        SPLITK_BYTES, stream()))
    ref_w, ref_b = _wgrad_ref(a.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2))
    if accumulate:
        ref_w, ref_b = ref_w + dw0.double(), ref_b + db0.double()
    assert (dw.double() - ref_w).abs().max().item() < 2e-6 * ref_w.abs().max().item()
    assert (db.double() - ref_b).abs().max().item() < 2e-6 * ref_b.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("N,C1,C2,Cout,H", [(4, 128, 128, 128, 32), (2, 256, 128, 256, 16), (3, 512, 384, 512, 8), (2, 128, 256, 128, 64)])
def test_resblock_training_node_two_sources(N, C1, C2, Cout, H):
    """ops.resblock_train on a CatAct: the skip concatenation th.cat([h, hs.pop()], dim=1) (unet.py:628) read in place by the first
    GroupNorm and the 1x1 skip conv, its gradient written to the two tensors separately — against autograd in fp64 on a real cat."""
    import torch.nn.functional as F
    from causaldiffae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(17)
    cl = torch.channels_last
    C = C1 + C2

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, device=dev, generator=g) * scale

    a0, b0 = rnd(N, C1, H, H).contiguous(memory_format=cl), rnd(N, C2, H, H).contiguous(memory_format=cl)
    ss0 = rnd(N, 2 * Cout, scale=0.2)
    p0 = dict(g1=1 + rnd(C, scale=0.1), b1=rnd(C, scale=0.1), w1=(rnd(Cout, C, 3, 3) / (3 * C ** 0.5)).contiguous(memory_format=cl),
              c1b=rnd(Cout, scale=0.1), g2=1 + rnd(Cout, scale=0.1), b2=rnd(Cout, scale=0.1),
              w2=(rnd(Cout, Cout, 3, 3) / (3 * Cout ** 0.5)).contiguous(memory_format=cl), c2b=rnd(Cout, scale=0.1),
              sw=rnd(Cout, C, 1, 1) / C ** 0.5, sb=rnd(Cout, scale=0.1))
    dy = rnd(N, Cout, H, H).contiguous(memory_format=cl) * 1e-3
    res = []
    with torch.enable_grad():
        for mode in ("f64", "node"):
            dt = torch.float64 if mode == "f64" else torch.float32
            a, b, ss = (t.detach().to(dt).requires_grad_() for t in (a0, b0, ss0))
            p = {k: v.detach().to(dt).requires_grad_() for k, v in p0.items()}
            if mode == "f64":
                x = torch.cat([a, b], dim=1)
                h = F.conv2d(F.silu(F.group_norm(x, 32, p["g1"], p["b1"], 1e-5)), p["w1"], p["c1b"], padding=1)
                h = F.group_norm(h, 32, p["g2"], p["b2"], 1e-5) * (1 + ss[:, :Cout, None, None]) + ss[:, Cout:, None, None]
                out = F.conv2d(F.silu(h), p["w2"], p["c2b"], padding=1) + F.conv2d(x, p["sw"], p["sb"])
            else:
                x = ops.cat_channels(a, b)
                assert isinstance(x, ops.CatAct)
                out = ops.resblock_train(x, ss, p["g1"], p["b1"], p["w1"], p["c1b"], p["g2"], p["b2"], p["w2"], p["c2b"], p["sw"], p["sb"])
            out.backward(dy.to(dt))
            res.append([out.detach(), a.grad, b.grad, ss.grad] + [p[k].grad for k in sorted(p)])
    names = ["out", "da", "db", "dss"] + sorted(p0)
    for n, r, f in zip(names, *res):
        assert (f.double().reshape(r.shape) - r).abs().max().item() < 3e-5 * r.abs().max().item(), n


# (N, C, Cout, H) of the C64 training benchmark at batch 32 per GPU (bench.py --train-batch 32) and of the sampling benchmark at batch
# 128: at these sizes the dispatcher takes convwin_kernel<f16, 9> for the forward and convwin_kernel<bf16, 9> for dgrad (split-K at
# the 16 x 16 and 8 x 8 levels) — the small cases above all run on the first-generation 128-row window kernel.
BENCH_NODE_SHAPES = [(32, 128, 128, 64), (32, 256, 256, 32), (64, 384, 384, 16), (128, 512, 512, 8)]


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,Cout,H", BENCH_NODE_SHAPES + [(32, 256, 128, 64)])
def test_fused_gn_conv_training_node_on_benchmark_dispatch(N, C, Cout, H, expect_kernels):
    """ops.gn_conv3x3 at the benchmark's sizes: forward on convwin_kernel<f16>, dgrad on convwin_kernel<bf16> (asserted from the
    library's launch log), wgrad on wgwin_kernel — against torch autograd in fp64 at the bar of the small cases (2e-5 of each
    tensor's own scale).  Reference ops: unet.py:185-198 forward / backward."""
    from causaldiffae_amd import ops
    with torch.enable_grad():
        assert ops.train_presplit_ok((N, C, H, H), Cout)
        with expect_kernels(convwin=1, convwin_dgrad=1):
            fused = _gnconv_case(N, C, Cout, H, True, True, "fused")
        ref = _gnconv_case(N, C, Cout, H, True, True, "f64")
    for name, f, r in zip(("out", "dx", "dgamma", "dbeta", "dss", "dw", "db", "dres"), fused, ref):
        sc = r.abs().max().item() + 1e-30
        assert (f.double() - r).abs().max().item() / sc < 2e-5, name


def _resblock_case(N, C1, C2, Cout, H, mode, seed=9):
    """One ResBlock (identity / 1x1 skip; C2 > 0: the input is the skip concatenation [a | b]) forward + backward; mode "f64" = torch
    autograd in fp64, "node" = ops.resblock_train.  Returns (names, tensors)."""
    import torch.nn.functional as F
    from causaldiffae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(seed)
    cl = torch.channels_last
    C = C1 + C2

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, device=dev, generator=g) * scale

    a0 = rnd(N, C1, H, H).contiguous(memory_format=cl)
    b0 = rnd(N, C2, H, H).contiguous(memory_format=cl) if C2 else None
    ss0 = rnd(N, 2 * Cout, scale=0.2)
    p0 = dict(g1=1 + rnd(C, scale=0.1), b1=rnd(C, scale=0.1), w1=(rnd(Cout, C, 3, 3) / (3 * C ** 0.5)).contiguous(memory_format=cl),
              c1b=rnd(Cout, scale=0.1), g2=1 + rnd(Cout, scale=0.1), b2=rnd(Cout, scale=0.1),
              w2=(rnd(Cout, Cout, 3, 3) / (3 * Cout ** 0.5)).contiguous(memory_format=cl), c2b=rnd(Cout, scale=0.1))
    if C != Cout:
        p0.update(sw=rnd(Cout, C, 1, 1) / C ** 0.5, sb=rnd(Cout, scale=0.1))
    dy = rnd(N, Cout, H, H).contiguous(memory_format=cl) * 1e-3
    dt = torch.float64 if mode == "f64" else torch.float32
    a, ss = a0.detach().to(dt).requires_grad_(), ss0.detach().to(dt).requires_grad_()
    b = b0.detach().to(dt).requires_grad_() if C2 else None
    p = {k: v.detach().to(dt).requires_grad_() for k, v in p0.items()}
    if mode == "f64":
        x = torch.cat([a, b], dim=1) if C2 else a
        h = F.conv2d(F.silu(F.group_norm(x, 32, p["g1"], p["b1"], 1e-5)), p["w1"], p["c1b"], padding=1)
        h = F.group_norm(h, 32, p["g2"], p["b2"], 1e-5) * (1 + ss[:, :Cout, None, None]) + ss[:, Cout:, None, None]
        out = F.conv2d(F.silu(h), p["w2"], p["c2b"], padding=1) + (x if C == Cout else F.conv2d(x, p["sw"], p["sb"]))
    else:
        x = ops.cat_channels(a, b) if C2 else a
        out = ops.resblock_train(x, ss, p["g1"], p["b1"], p["w1"], p["c1b"], p["g2"], p["b2"], p["w2"], p["c2b"], p.get("sw"), p.get("sb"))
    out.backward(dy.to(dt))
    names = ["out", "da"] + (["db"] if C2 else []) + ["dss"] + sorted(p0)
    return names, [out.detach(), a.grad] + ([b.grad] if C2 else []) + [ss.grad] + [p[k].grad for k in sorted(p)]


@pytest.mark.gpu
@pytest.mark.parametrize("N,C1,C2,Cout,H", [(32, 128, 0, 128, 64), (32, 256, 0, 256, 32), (64, 384, 0, 384, 16), (128, 512, 0, 512, 8),
                                             (32, 128, 0, 256, 32),        # 1x1 skip conv
                                             (32, 256, 128, 128, 64),      # skip concatenation at the 64 x 64 level (last up block of the C64 UNet)
                                             (32, 512, 384, 384, 16)])     # skip concatenation, split-K dgrad
def test_resblock_training_node_on_benchmark_dispatch(N, C1, C2, Cout, H, expect_kernels):
    """ops.resblock_train (the node bench.py's training leg spends its time in) at the benchmark's per-GPU sizes, incl. a 1x1-skip and
    two two-source blocks: both convs' forward on convwin_kernel<f16>, both dgrads on convwin_kernel<bf16> (asserted from the launch
    log), against torch autograd in fp64 for every input and parameter at the small cases' bar (3e-5 of each tensor's own scale)."""
    with torch.enable_grad():
        with expect_kernels(convwin=2, convwin_dgrad=2):
            names, got = _resblock_case(N, C1, C2, Cout, H, "node")
        _, ref = _resblock_case(N, C1, C2, Cout, H, "f64")
    for n, f, r in zip(names, got, ref):
        assert (f.double().reshape(r.shape) - r).abs().max().item() < 3e-5 * r.abs().max().item(), n


@pytest.mark.gpu
@pytest.mark.parametrize("N,C1,C2,Cout,H", [(2, 128, 0, 128, 64), (2, 256, 128, 128, 32), (3, 512, 0, 512, 8), (2, 128, 0, 256, 16)])
def test_resblock_training_node_forced_onto_window_kernel(N, C1, C2, Cout, H, expect_kernels):
    """The small shapes of the model-level goldens (N = 2) pushed through convwin_kernel forward and dgrad by the dispatch threshold
    (cdae_tune_set CDAE_TUNE_CONVWIN_MIN_TILES = 1): partial tiles, one tile per block, split-K slabs of a few rows."""
    from causaldiffae_amd._lib import tune_scope
    with torch.enable_grad():
        with tune_scope(convwin_min_tiles=1), expect_kernels(convwin=2, convwin_dgrad=2):
            names, got = _resblock_case(N, C1, C2, Cout, H, "node", seed=23)
        _, ref = _resblock_case(N, C1, C2, Cout, H, "f64", seed=23)
    for n, f, r in zip(names, got, ref):
        assert (f.double().reshape(r.shape) - r).abs().max().item() < 3e-5 * r.abs().max().item(), n



@pytest.mark.gpu
@pytest.mark.parametrize("N,C1,C2,Cout,H", [(32, 128, 0, 128, 64), (32, 256, 128, 256, 32), (4, 128, 0, 256, 16), (64, 512, 0, 512, 8)])
def test_resblock_training_node_mixed16_single_plane_kernels(N, C1, C2, Cout, H, expect_kernels):
    """The reduced-precision torso (`mixed16`, reference unet.py:501-507 / fp16_util.py:9-15: an fp16 torso with fp32 master weights) on the
    fused ResBlock node: ONE f16 plane per operand in the forward convs, ONE bf16 plane in dgrad and wgrad (convwin_kernel<.., 1 plane>,
    wgwin_kernel<1 plane>: one MFMA per product).  Against fp64 at the accuracy a 2^-8 (bf16) / 2^-11 (f16) significand gives — and
    well away from the f16x3 path's 3e-5, i.e. the single-plane kernels really ran."""
    from causaldiffae_amd._lib import precision_scope
    with torch.enable_grad():
        with precision_scope("mixed16"), expect_kernels(**({"convwin": 2, "convwin_dgrad": 2} if N >= 32 else {})):
            names, got = _resblock_case(N, C1, C2, Cout, H, "node", seed=31)
        _, ref = _resblock_case(N, C1, C2, Cout, H, "f64", seed=31)
    worst = 0.0
    for n, f, r in zip(names, got, ref):
        e = (f.double().reshape(r.shape) - r).abs().max().item() / r.abs().max().item()
        assert e < (4e-3 if n == "out" else 3e-2), (n, e)
        worst = max(worst, e)
    assert worst > 1e-4, worst


@pytest.mark.gpu
@pytest.mark.parametrize("stream_kernel", [True, False])
@pytest.mark.parametrize("N,C1,C2,Cout,H", [(4, 128, 128, 128, 32), (2, 256, 128, 256, 16), (3, 512, 384, 512, 8), (2, 128, 0, 256, 32),
                                            (1, 64, 64, 96, 16), (6, 256, 0, 384, 4), (3, 128, 128, 384, 8), (1, 128, 128, 128, 64)])
def test_skip_gemm_carries_groupnorm_planes(N, C1, C2, Cout, H, stream_kernel, monkeypatch):
    """The ResBlock entry sweep (cdae_skip_gn_fwd, and the igemm-loader version cdae_linear_fwd_cat_gn it replaced): the 1x1 skip conv
    that also writes GroupNorm+SiLU of the (concatenated) block input as f16 planes — the planes bit-identical to
    cdae_gn_apply_split2's, the GEMM result equal to cdae_linear_fwd_cat's and to fp64.  Shapes: partial row tiles (M = 192, 80),
    several images per 128-row tile, a partial column tile (Cout = 96), 3 column tiles, a table too large for LDS (falls back)."""
    from causaldiffae_amd import ops
    monkeypatch.setattr(ops, "_SKIPGN_V2", stream_kernel)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(23)
    cl = torch.channels_last
    C = C1 + C2
    a = torch.randn(N, C1, H, H, device=dev, generator=g).contiguous(memory_format=cl)
    b = torch.randn(N, C2, H, H, device=dev, generator=g).contiguous(memory_format=cl) if C2 else None
    gamma = 1 + 0.1 * torch.randn(C, device=dev, generator=g)
    beta = 0.1 * torch.randn(C, device=dev, generator=g)
    w = torch.randn(Cout, C, 1, 1, device=dev, generator=g) / C ** 0.5
    bias = 0.1 * torch.randn(Cout, device=dev, generator=g)
    with torch.no_grad():
        x = ops.CatAct(a, b) if C2 else a
        lz = ops.group_norm_lazy(x, gamma, beta, None, True, 32, 1e-5)
        assert ops.skip_gn_ok(lz, w)
        skip, planes = ops.skip_gn_fused(lz, w, bias)
        ref_planes = lz.planes()
        ref_skip = ops.conv1x1_cat(x, w.reshape(Cout, C), bias) if C2 else ops.conv1x1(a, w.reshape(Cout, C), bias)
    assert torch.equal(planes.hi, ref_planes.hi) and torch.equal(planes.lo, ref_planes.lo)
    # the GEMM itself: same products, but the separate launch may split K on these small grids (different fp32 summation order)
    assert (skip - ref_skip).abs().max().item() < 2e-5 * ref_skip.abs().max().item()
    xx = torch.cat([a, b], dim=1) if C2 else a
    ref64 = torch.einsum("nchw,oc->nohw", xx.double(), w.reshape(Cout, C).double()) + bias.double()[None, :, None, None]
    assert (skip.double() - ref64).abs().max().item() < 4e-6 * ref64.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_skip_sweep_random_shapes(seed):
    """Seeded fuzz of the ResBlock entry sweep (skipgn_kernel): one or two sources, channel counts in steps of 32, partial row and column
    tiles, pixel-major and group-major plane output, with the coefficient table taken from the statistics launch — planes bit-identical
    to the GroupNorm apply kernel's, the 1x1 GEMM against fp64."""
    import random
    from causaldiffae_amd import ops
    rng = random.Random(5000 + seed)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(61 + seed)
    cl = torch.channels_last
    done = 0
    while done < 6:
        H = rng.choice([8, 16, 32])
        N = rng.randint(1, 5 if H == 32 else 20)
        C1, C2, Cout = 32 * rng.randint(1, 8), 32 * rng.randint(0, 6), 32 * rng.randint(2, 12)
        C = C1 + C2
        if (C // 32) % 4:                         # the plane kernels want 4-channel vectors inside a group
            continue
        a = torch.randn(N, C1, H, H, device=dev, generator=g).contiguous(memory_format=cl)
        b = torch.randn(N, C2, H, H, device=dev, generator=g).contiguous(memory_format=cl) if C2 else None
        gamma, beta = 1 + 0.1 * torch.randn(C, device=dev, generator=g), 0.1 * torch.randn(C, device=dev, generator=g)
        w = torch.randn(Cout, C, 1, 1, device=dev, generator=g) / C ** 0.5
        bias = 0.1 * torch.randn(Cout, device=dev, generator=g)
        gm = rng.random() < 0.5
        with torch.no_grad():
            x = ops.CatAct(a, b) if C2 else a
            lz = ops.group_norm_lazy(x, gamma, beta, None, True, 32, 1e-5, want_coef=True)
            if not ops.skip_gn_ok(lz, w):
                continue
            skip, planes = ops.skip_gn_fused(lz, w, bias, gm=gm)
            ref_planes = lz.planes(gm=gm)
        assert planes.gm == ref_planes.gm and torch.equal(planes.hi, ref_planes.hi) and torch.equal(planes.lo, ref_planes.lo), (N, C1, C2, Cout, H, gm)
        xx = torch.cat([a, b], dim=1) if C2 else a
        ref64 = torch.einsum("nchw,oc->nohw", xx.double(), w.reshape(Cout, C).double()) + bias.double()[None, :, None, None]
        assert (skip.double() - ref64).abs().max().item() < 4e-6 * ref64.abs().max().item(), (N, C1, C2, Cout, H, gm)
        done += 1


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,Nf,H,silu", [(32, 384, 1152, 16, False), (64, 512, 1536, 8, False), (16, 256, 768, 16, True), (5, 384, 1152, 32, False),
                                           (33, 128, 96, 12, True)])
def test_groupnorm_linear_in_one_pass(N, C, Nf, H, silu):
    """ops.linear_gn (cdae_linear_fwd_stream_gn: GroupNorm folded into the streaming GEMM's row loader — the attention block's
    norm -> qkv) against the two-pass path (GroupNorm planes, then the plane GEMM: same operand values, other summation order) and fp64.
    Shapes: both attention levels of the P64 / C64 UNets, several images per 128-row tile, a partial row tile and column tile."""
    import torch.nn.functional as F
    from causaldiffae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(29)
    x = (torch.randn(N, C, H, H, device=dev, generator=g) * 1.3 + 0.2).contiguous(memory_format=torch.channels_last)
    gamma = 1 + 0.1 * torch.randn(C, device=dev, generator=g)
    beta = 0.1 * torch.randn(C, device=dev, generator=g)
    w = torch.randn(Nf, C, device=dev, generator=g) / C ** 0.5
    bias = 0.1 * torch.randn(Nf, device=dev, generator=g)
    with torch.no_grad():
        lz = ops.group_norm_lazy(x, gamma, beta, None, silu, 32, 1e-5)
        assert ops.linear_gn_ok(lz, w)
        got = ops.linear_gn(lz, w, bias)
        two = ops.linear_ps(lz.planes(), w, bias)
    h = F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5)
    if silu:
        h = F.silu(h)
    ref = h.permute(0, 2, 3, 1).reshape(N * H * H, C) @ w.double().t() + bias.double()
    scale = ref.abs().max().item()
    assert (got - two).abs().max().item() < 2e-5 * scale
    assert (got.double() - ref).abs().max().item() < 6e-6 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,Cout,H,W,silu", [(3, 128, 4, 64, 64, True), (2, 128, 3, 64, 64, True), (5, 128, 1, 32, 32, True), (2, 128, 8, 28, 28, True),
                                               (2, 128, 6, 9, 7, False), (1, 128, 2, 8, 12, True)])
@pytest.mark.parametrize("mfma", [1, 0])
def test_output_head_kernel_matches_fp64(N, C, Cout, H, W, silu, mfma):
    """cdae_head_conv_fwd (GroupNorm -> SiLU -> conv3x3 to a few channels in one exact-fp32 kernel, the UNet's self.out) against
    group_norm + conv2d in fp64: every supported channel count, image sizes that do not fill the 256-pixel tile, odd widths — on the
    4 x 4 x 1 matrix-core form (the default) and on the scalar form it replaced (CDAE_TUNE_HEAD_MFMA = 0)."""
    import torch.nn.functional as F
    from causaldiffae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(41)
    x = ops.to_nhwc(torch.randn(N, C, H, W, device=dev, generator=g) * 1.7 + 0.3)
    gamma, beta = 1 + 0.2 * torch.randn(C, device=dev, generator=g), 0.2 * torch.randn(C, device=dev, generator=g)
    w = (torch.randn(Cout, C, 3, 3, device=dev, generator=g) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
    b = 0.1 * torch.randn(Cout, device=dev, generator=g)
    with torch.no_grad():
        lz = ops.group_norm_lazy(x, gamma, beta, None, silu, 32, 1e-5)
        assert ops.head_conv_ok(lz, w)
        from causaldiffae_amd._lib import tune_scope
        with tune_scope(head_mfma=mfma):
            got = ops.head_conv(lz, w, b)
        planes = ops.conv3x3_ps(lz.planes(), w, b, out_nchw=True)           # the path it replaces
    h = F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5)
    ref = F.conv2d(F.silu(h) if silu else h, w.double(), b.double(), padding=1)
    assert got.shape == ref.shape and got.is_contiguous()
    scale = ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() < 2e-6 * scale
    assert (planes.double() - ref).abs().max().item() < 1e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,H,W,nchw", [(3, 4, 128, 16, 16, True), (2, 3, 128, 64, 64, True), (5, 1, 128, 28, 28, True), (2, 4, 64, 8, 12, False),
                                                 (2, 2, 96, 9, 7, True)])
def test_stem_conv_matches_fp64(N, Cin, Cout, H, W, nchw):
    """cdae_conv3x3_stem (the UNet input conv, 1..4 channels, exact fp32 on the vector ALUs) through ops.conv3x3 — NCHW model inputs and
    NHWC tensors, odd image sizes, channel-group counts that do not divide the block — against conv2d in fp64."""
    import torch.nn.functional as F
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import lib
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(31)
    x = torch.randn(N, Cin, H, W, device=dev, generator=g)
    if not nchw:
        x = x.contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / (3 * Cin ** 0.5)).contiguous(memory_format=torch.channels_last)
    b = 0.1 * torch.randn(Cout, device=dev, generator=g)
    assert lib.cdae_conv3x3_stem_supported(Cin, Cout, W) == 1
    with torch.no_grad():
        got = ops.conv3x3(x, w, b)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert got.shape == ref.shape
    assert (got.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("causal,nv", [(True, 4), (True, 2), (False, 4)])
def test_representation_loss_kernel(causal, nv):
    """cdae_rep_loss / _bwd (one launch each way) against the reference's chain of element-wise ops (gaussian_diffusion.py:727-766 with
    nn.py:440-457: kl_normal(mu, var, 0, 1) + sum_i kl_normal(z_post_i, 1, c_i, 1)) in fp64: values and the gradients with respect to
    mu, var and z_post; and GaussianDiffusion.representation_loss takes that path (masked and unmasked)."""
    from causaldiffae_amd import ops
    from causaldiffae_amd.nn import kl_normal
    from improved_diffusion import script_util as su
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(3)
    N, D = 7, 512
    mu = torch.randn(N, D, device=dev, generator=g, requires_grad=True)
    var = (torch.rand(N, D, device=dev, generator=g) * 2 + 1e-3).requires_grad_(True)
    zp = torch.randn(N, D, device=dev, generator=g, requires_grad=True)
    c = torch.rand(N, nv, device=dev, generator=g)
    w = torch.randn(N, device=dev, generator=g)
    out = ops.rep_loss(mu, var, zp if causal else None, c if causal else None)
    (out * w).sum().backward()
    mu64, var64, zp64 = (t.detach().double().requires_grad_(True) for t in (mu, var, zp))
    ref = kl_normal(mu64, var64, torch.zeros_like(mu64), torch.ones_like(var64))
    if causal:
        d = D // nv
        for i in range(nv):
            zi = zp64.reshape(N, nv, d)[:, i]
            ref = ref + kl_normal(zi, torch.ones_like(zi), c.double()[:, i:i + 1].expand(-1, d), torch.ones_like(zi))
    (ref * w.double()).sum().backward()
    assert (out.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    for got, want in ((mu.grad, mu64.grad), (var.grad, var64.grad)) + (((zp.grad, zp64.grad),) if causal else ()):
        assert (got.double() - want).abs().max().item() < 2e-6 * want.abs().max().item()
    if not causal:
        assert zp.grad is None
    _, diff = su.create_model_and_diffusion(**{**su.model_and_diffusion_defaults(), "image_size": 32, "in_channels": 1, "n_vars": nv, "rep_cond": True,
                                              "causal_modeling": causal, "num_channels": 32, "num_res_blocks": 1})
    with torch.no_grad():
        k1 = diff.representation_loss(mu, var, zp, causal, None, c)
        mask = (torch.arange(N, device=dev) % 2).float()
        k2 = diff.representation_loss(mu, var, zp, causal, mask, c)
    assert (k1.double() - ref.detach()).abs().max().item() < 2e-6 * ref.abs().max().item()
    assert abs(k2.item() - ((ref.detach() * mask.double()).sum() / mask.sum()).item()) < 2e-6 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("M,C,HW", [(8192, 384, 256), (4096 + 64, 512, 64)])
def test_streaming_gemm_leaves_groupnorm_sums(M, C, HW):
    """cdae_linear_fwd_stream_gn_part (the attention block's proj_out + residual): result bit-identical to the streaming GEMM without
    sums, per (32-row chunk, column) sums equal to those of the result rows — also with a partial last row tile."""
    from causaldiffae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(29)
    rows = torch.randn(M, C, device=dev, generator=g)
    res = torch.randn(M, C, device=dev, generator=g)
    w = torch.randn(C, C, device=dev, generator=g) / C ** 0.5
    b = torch.randn(C, device=dev, generator=g)
    with torch.no_grad():
        hit = ops.linear_stream_gn(rows, w, b, res, HW)
        assert hit is not None
        y, parts = hit
        plain = ops.linear(rows, w, b, res=res)
    assert torch.equal(y, plain)
    exact = rows.double() @ w.double().t() + b.double() + res.double()
    assert (y.double() - exact).abs().max().item() < 6e-6 * exact.abs().max().item()
    r3 = y.double().reshape(M // 32, 32, C)
    assert parts.shape == (M // 32, C, 2)
    assert (parts[:, :, 0].double() - r3.sum(1)).abs().max().item() < 1e-5 * r3.sum(1).abs().max().item()
    assert (parts[:, :, 1].double() - (r3 * r3).sum(1)).abs().max().item() < 1e-5 * (r3 * r3).sum(1).abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,S", [(3, 4, 128, 64), (5, 1, 128, 32), (2, 3, 256, 64), (2, 4, 128, 128)])
def test_stem_conv_leaves_groupnorm_sums(N, Cin, Cout, S):
    """cdae_conv3x3_stem_gn: the input conv's result (bit-identical to the plain stem kernel) with per (32-pixel chunk, channel) sums and
    sums of squares attached; a GroupNorm that takes its statistics from them equals one that runs its own statistics pass."""
    from causaldiffae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(17)
    x = torch.randn(N, Cin, S, S, device=dev, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / (3 * Cin ** 0.5)).contiguous(memory_format=torch.channels_last)
    b = 0.1 * torch.randn(Cout, device=dev, generator=g)
    with torch.no_grad():
        assert ops.stem_conv_gn_ok(x, w)
        y = ops.stem_conv_gn(x, w, b)
        plain = ops.conv3x3(x, w, b)
        assert torch.equal(y, plain)
        rows = y.permute(0, 2, 3, 1).reshape(-1, 32, Cout).double()
        parts = y._gnparts.double()
        assert parts.shape == (N * S * S // 32, Cout, 2)
        assert (parts[:, :, 0] - rows.sum(1)).abs().max().item() < 1e-5 * rows.sum(1).abs().max().item()
        assert (parts[:, :, 1] - (rows * rows).sum(1)).abs().max().item() < 1e-5 * (rows * rows).sum(1).abs().max().item()
        gamma, beta = torch.rand(Cout, device=dev, generator=g) + 0.5, torch.randn(Cout, device=dev, generator=g)
        got = ops.group_norm_split(y, gamma, beta, None, True)
        ref = ops.group_norm_split(plain, gamma, beta, None, True)
    d = ((got.hi.float() + got.lo.float()) - (ref.hi.float() + ref.lo.float())).abs().max().item()
    assert d < 2e-5, d


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,k0,kc", [(8192, 384, 384, 0, None), (131072, 128, 256, 0, 128), (131072, 128, 256, 128, 128), (4096 + 96, 1152, 384, 0, None),
                                          (32768, 256, 640, 384, 256)])
def test_linear_dgrad_streaming_bf16(M, N, K, k0, kc):
    """dx = dy @ W[:, k0 : k0 + kc] on the streaming kernel (skipgn_kernel<false, bf16>: dy split to bf16 hi / lo on its way into LDS,
    bf16 planes of W^T from cdae_wt_planes_bf16) against fp64 at the gradient bar (2e-4 of the gradient's maximum) and against the tiled
    fp32-operand GEMM it replaces for large M; gradient-sized dy (1e-4 ... 1e-9: bf16 keeps fp32's range), partial row tile, column range."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import precision_scope
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(83)
    dy = torch.randn(M, N, device=dev, generator=g) * torch.logspace(-4, -9, N, device=dev)[None, :]
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    cols = K if kc is None else kc
    with precision_scope("f16x3"), torch.no_grad():
        dx = torch.empty(M, cols, device=dev)
        ops.linear_dgrad(dy, N, w, dx, cols, M, N, K, k0, kc)
        ops._DGRAD_STREAM_ON = False
        try:
            dx_old = torch.empty(M, cols, device=dev)
            ops.linear_dgrad(dy, N, w, dx_old, cols, M, N, K, k0, kc)
        finally:
            ops._DGRAD_STREAM_ON = True
    ref = dy.double() @ w.double()[:, k0:k0 + cols]
    scale = ref.abs().max().item()
    assert torch.isfinite(dx).all()
    assert (dx.double() - ref).abs().max().item() < 2e-4 * scale, (dx.double() - ref).abs().max().item() / scale
    assert (dx_old.double() - ref).abs().max().item() < 2e-4 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,S", [(32, 128, 128, 64), (8, 256, 256, 32), (4, 384, 384, 16), (2, 64, 96, 16), (3, 32, 64, 32)])
def test_stride2_conv_dgrad_as_subpixel_phases(N, Cin, Cout, S, expect_kernels):
    """dgrad of the Downsample conv (stride 2, pad 1: reference unet.py:82-105) as four 2 x 2 sub-pixel convolutions of dy on the plane
    kernels (cdae_conv3x3_s2_dgrad_ps: the window kernel's 4-tap bf16 instantiation where the grid fills the chip, the small-grid plane
    kernels phase by phase elsewhere) against conv2d's autograd in fp64 at the gradient bar (2e-4 of the gradient's maximum; bf16x3 =
    2^-16 products), and against the masked 9-tap gather it replaces."""
    from causaldiffae_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(61)
    x = ops.to_nhwc(torch.randn(N, Cin, S, S, device="cuda:0", generator=g)).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda:0", generator=g) / (9 * Cin) ** 0.5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(N, Cout, S // 2, S // 2, device="cuda:0", generator=g) * 1e-3          # gradient-sized values
    from causaldiffae_amd._lib import precision_scope
    with precision_scope("f16x3"):
        y = ops.conv3x3(x, w, None, stride=2)
        want = dict(convwin_up=1) if N * (S // 2) ** 2 // 256 * ((Cin + 127) // 128) * 4 >= 256 and (Cout // 32) < 6 else {}
        with expect_kernels(**want):
            (dx_new,) = torch.autograd.grad(y, x, gy, retain_graph=True)
        ops._S2DGRAD_ON = False
        try:
            (dx_old,) = torch.autograd.grad(y, x, gy, retain_graph=True)
        finally:
            ops._S2DGRAD_ON = True
    xd, wd = x.detach().double().contiguous().requires_grad_(True), w.detach().double()
    (dx_ref,) = torch.autograd.grad(F.conv2d(xd, wd, stride=2, padding=1), xd, gy.double())
    scale = dx_ref.abs().max().item()
    assert torch.isfinite(dx_new).all()
    assert (dx_new.double() - dx_ref).abs().max().item() < 2e-4 * scale, (dx_new.double() - dx_ref).abs().max().item() / scale
    assert (dx_old.double() - dx_ref).abs().max().item() < 2e-4 * scale


# ------------------------------------------------------------------ dynamic range of the split-precision (f16x3) contractions
def _rel(got, exact):
    return (got.double().cpu() - exact.cpu()).abs().max().item() / exact.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("e", [-16, -12, -8, 0, 8, 12])
def test_split_precision_dynamic_range(e):
    """conv3x3 / linear on pre-split planes and the fused attention with one operand scaled by 2^e, against fp64.
    WEIGHTS are scale invariant: their planes hold w * 2^k with k per tensor (include/cdae.h, cdae_weight_scales), so the error is that
    of the split (2^-22 relative) for every weight scale — bound 6e-6.  ACTIVATION planes are unscaled (the network's activations are
    normalised): below 2^-3 their lo plane is an f16 subnormal and the pair turns into fixed point with an LSB of 2^-24 — the
    documented bound there is 4 * 2^-24 / max|operand| relative to the result."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import get_precision, range_check, set_precision
    prev = get_precision()
    set_precision("f16x3")
    try:
        range_check("stale")
    except Exception:
        pass
    try:
        def bound_for(t):          # activations: 2^-22-relative in range; fixed point with an LSB of 2^-24 once the lo plane is subnormal
            return max(6e-6, 4.0 * 2.0 ** -24 / t.abs().max().item())
        g = torch.Generator(device="cuda:0").manual_seed(21)
        sc = 2.0 ** e
        # conv3x3 on planes: scaled activation, fan-in-scaled weights
        x = ops.to_nhwc(torch.randn(4, 128, 16, 16, device="cuda:0", generator=g) * sc)
        w = (torch.randn(128, 128, 3, 3, device="cuda:0", generator=g) / 1152 ** 0.5).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            y = ops.conv3x3_ps(_split_nhwc(x), w, None)
        exact = F.conv2d(x.double().contiguous(), w.double(), padding=1)
        assert torch.isfinite(y).all() and _rel(y, exact) < bound_for(x), (e, _rel(y, exact))
        # the same conv with the WEIGHTS scaled instead: 6e-6 whatever the scale
        w2 = (w * sc).contiguous(memory_format=torch.channels_last)
        x1 = ops.to_nhwc(torch.randn(4, 128, 16, 16, device="cuda:0", generator=g))
        with torch.no_grad():
            y = ops.conv3x3_ps(_split_nhwc(x1), w2, None)
        exact = F.conv2d(x1.double().contiguous(), w2.double(), padding=1)
        assert torch.isfinite(y).all() and _rel(y, exact) < 6e-6, (e, _rel(y, exact))
        # ... through the fused sub-pixel up-conv (folded weights, a scale record of the folded tensor) ...
        with torch.no_grad():
            yu = ops.conv3x3_ps(_split_nhwc(x1), w2, None, up=True)
        exact = F.conv2d(F.interpolate(x1.double().contiguous(), scale_factor=2, mode="nearest"), w2.double(), padding=1)
        assert torch.isfinite(yu).all() and _rel(yu, exact) < 6e-6, (e, _rel(yu, exact))
        # ... and through the fp32-operand kernel (the weight tile is scaled before the in-kernel split), stride 2
        with torch.no_grad():
            ys = ops.conv3x3(x1, w2, None, stride=2)
        exact = F.conv2d(x1.double().contiguous(), w2.double(), padding=1, stride=2)
        assert torch.isfinite(ys).all() and _rel(ys, exact) < 6e-6, (e, _rel(ys, exact))
        # linear on planes: scaled activation
        xl = ops.to_nhwc(torch.randn(2, 256, 8, 8, device="cuda:0", generator=g) * sc)
        wl = torch.randn(384, 256, 1, device="cuda:0", generator=g) / 16.0
        with torch.no_grad():
            yl = ops.linear_ps(_split_nhwc(xl), wl, None)
        exact = xl.permute(0, 2, 3, 1).reshape(-1, 256).double() @ wl[:, :, 0].double().t()
        assert torch.isfinite(yl).all() and _rel(yl, exact) < bound_for(xl), (e, _rel(yl, exact))
        # linear with the WEIGHT scaled: the plane GEMM, the streaming GEMM (M >= 4096 rows) and the fp32-operand kernel
        wl2 = (wl * sc).contiguous()
        xr = torch.randn(8192, 256, device="cuda:0", generator=g)
        exact = xr.double() @ wl2[:, :, 0].double().t()
        with torch.no_grad():
            y_ps = ops.linear_ps(_split_nhwc(xr.reshape(8, 32, 32, 256).permute(0, 3, 1, 2)), wl2, None)
            y_st = ops.linear(xr, wl2)                      # streaming GEMM on pre-split weight planes
            y_ig = ops.linear(xr[:512], wl2)                # below the streaming threshold: igemm, in-kernel split
        assert _rel(y_ps, exact) < 6e-6 and _rel(y_st, exact) < 6e-6 and _rel(y_ig, exact[:512]) < 6e-6, (e, _rel(y_ps, exact), _rel(y_st, exact), _rel(y_ig, exact[:512]))
        # fused attention, values scaled (q / k unscaled: the softmax is scale sensitive by definition)
        B, T, heads, ch = 2, 64, 2, 96
        qkv = torch.randn(B, T, 3 * heads * ch, device="cuda:0", generator=g)
        q5 = qkv.reshape(B, T, heads, 3, ch).clone()
        q5[:, :, :, 2] *= sc
        qkv = q5.reshape(B, T, 3 * heads * ch).contiguous()
        with torch.no_grad():
            a = ops.qkv_attention(qkv, heads)
        xd = qkv.double().reshape(B, T, heads, 3, ch)
        wgt = torch.softmax(torch.einsum("bthc,bshc->bhts", xd[:, :, :, 0], xd[:, :, :, 1]) / ch ** 0.5, dim=-1)
        exact = torch.einsum("bhts,bshc->bthc", wgt, xd[:, :, :, 2]).reshape(B, T, heads * ch)
        assert torch.isfinite(a).all() and _rel(a, exact) < max(bound_for(q5[:, :, :, 2]), 2e-5), (e, _rel(a, exact))
        range_check("in-range operands")          # nothing above may have raised the flag
    finally:
        set_precision(prev)


@pytest.mark.gpu
def test_weight_scale_records():
    """cdae_weight_scale1 / cdae_weight_scales: record {2^k, 2^-k} with k = 14 - floor(log2 max|w|), {1, 1} for an all-zero or
    non-finite tensor; the table form over several tensors of one flat buffer (chunked, with tensors longer and shorter than a chunk)
    agrees with the single-tensor form; planes of cdae_split_f16w hold w * 2^k and reconstruct w to 2^-22 relative or 2^-39 of the
    tensor maximum."""
    import math
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import check, lib, ptr, ptr2, stream
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    cases = [torch.randn(1000, device=dev, generator=g) * 0.02, torch.randn(40000, device=dev, generator=g) * 3e-6, torch.randn(7, device=dev, generator=g) * 5e4,
             torch.zeros(128, device=dev), torch.tensor([1.0, -2.0, 0.5, 0.25], device=dev), torch.tensor([1e-3, float("inf")] * 2, device=dev),
             torch.tensor([float("nan"), 1.0, 1.0, 1.0], device=dev), torch.full((64,), 2.0 ** -20, device=dev), torch.full((8,), 3e38, device=dev)]

    def expect(t):
        m = t.abs().max().item()
        if m == 0.0 or not math.isfinite(m):
            return 1.0, 1.0
        k = 14 - math.floor(math.log2(m))
        k = max(-126, min(126, k))
        return 2.0 ** k, 2.0 ** -k

    for t in cases:
        rec = ops.weight_scale(t)
        assert tuple(rec.tolist()) == expect(t), (t.flatten()[:4], rec.tolist(), expect(t))
    # the table form: tensors packed into one flat buffer at ARBITRARY element offsets (the max pass reads 16-byte vectors over the aligned
    # middle of a chunk and scalars at its ends)
    lens = [t.numel() for t in cases]
    offs, o = [], 1
    for k, n in enumerate(lens):
        offs.append(o)
        o += n + (k % 3)
    flat = torch.zeros(o, device=dev)
    views = []
    for t, of in zip(cases, offs):
        flat[of:of + t.numel()] = t
        views.append(flat[of:of + t.numel()])
    table = ops.ScaleTable(flat, views)
    table.refresh()
    assert [tuple(r) for r in table.records.tolist()] == [expect(t) for t in cases]
    flat[offs[0]] = 100.0                                     # a new weight version: records follow after the epoch bump
    ops.bump_weight_epoch()
    assert tuple(table.record(views[0]).tolist()) == (2.0 ** 8, 2.0 ** -8)
    # scaled planes reconstruct the weight
    for t in (cases[0], cases[1], cases[2][:4]):
        t = t[:t.numel() // 4 * 4].contiguous()
        rec = ops.weight_scale(t)
        planes = torch.empty((2, t.numel()), dtype=torch.float16, device=dev)
        check(lib.cdae_split_f16w(ptr(t), ptr(rec), *ptr2(planes), t.numel(), stream()))
        assert torch.isfinite(planes).all() and 2.0 ** 14 <= planes[0].float().abs().max().item() <= 2.0 ** 15
        back = (planes[0].double() + planes[1].double()) * rec[1].item()
        tol = torch.maximum(t.double().abs() * 2.0 ** -21, torch.full_like(t.double(), t.abs().max().item() * 2.0 ** -38))
        assert ((back - t.double()).abs() <= tol).all()


@pytest.mark.gpu
def test_trained_like_weight_distribution_conv():
    """A conv3x3 whose weights are log-uniform over four decades (oracle.closed_form.fill_value_trained) on every forward kernel family —
    window kernel (benchmark dispatch), small-grid plane kernel, fp32-operand kernel — against fp64 at the split's own bound, i.e. the
    same 6e-6 the uniform +-sqrt(3 / fan_in) weights get: no weight lands in a subnormal lo plane any more."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import precision_scope, tune_scope
    from oracle.closed_form import fill_value_trained
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(9)
    w = fill_value_trained("input_blocks.4.0.in_layers.2.weight", (256, 128, 3, 3)).to(dev).contiguous(memory_format=torch.channels_last)
    assert w.abs().min().item() < 2e-5 and w.abs().max().item() > 0.1
    x = ops.to_nhwc(torch.randn(2, 128, 32, 32, device=dev, generator=g))
    exact = F.conv2d(x.double().contiguous(), w.double(), padding=1)
    with precision_scope("f16x3"), torch.no_grad():
        y_small = ops.conv3x3_ps(_split_nhwc(x), w, None)
        with tune_scope(convwin_min_tiles=1):
            y_win = ops.conv3x3_ps(_split_nhwc(x), w, None)
        y_ig = ops.conv3x3(x, w, None)
    for name, y in (("plane", y_small), ("window", y_win), ("fp32-operand", y_ig)):
        assert _rel(y, exact) < 6e-6, (name, _rel(y, exact))


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["conv_act", "conv_weight", "linear", "attention"])
def test_split_precision_overflow_raises(which):
    """Activation operands beyond the f16 range (2^16-scaled normal data: |x| up to ~3e5 > 65504) must not come back as inf / NaN tensors
    unnoticed: the library's range flag is raised and the Python API turns it into CdaeRangeError; the `fp32` mode (IEEE fp32
    products, fp32 range) computes the same call correctly."""
    import causaldiffae_amd
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import CdaeRangeError, get_precision, range_check, set_precision
    prev = get_precision()
    g = torch.Generator(device="cuda:0").manual_seed(22)
    sc = 2.0 ** 16

    def run():
        with torch.no_grad():
            if which in ("conv_act", "conv_weight"):
                x = ops.to_nhwc(torch.randn(2, 64, 16, 16, device="cuda:0", generator=g) * (sc if which == "conv_act" else 1.0))
                w = (torch.randn(64, 64, 3, 3, device="cuda:0", generator=g) / 24.0 * (sc * 64 if which == "conv_weight" else 1.0)).contiguous(memory_format=torch.channels_last)
                y = ops.conv3x3_ps(_split_nhwc(x), w, None) if get_precision() != "fp32" else ops.conv3x3(x, w, None)
                return y, F.conv2d(x.double().contiguous(), w.double(), padding=1)
            if which == "linear":
                x = torch.randn(256, 128, device="cuda:0", generator=g) * sc
                w = torch.randn(64, 128, device="cuda:0", generator=g) / 11.0
                return ops.linear(x, w, None), x.double() @ w.double().t()
            B, T, heads, ch = 1, 64, 1, 128
            qkv = torch.randn(B, T, 3 * ch, device="cuda:0", generator=g)
            qkv[:, :, 2 * ch:] *= sc
            xd = qkv.double().reshape(B, T, heads, 3, ch)
            wgt = torch.softmax(torch.einsum("bthc,bshc->bhts", xd[:, :, :, 0], xd[:, :, :, 1]) / ch ** 0.5, dim=-1)
            return ops.qkv_attention(qkv, heads), torch.einsum("bhts,bshc->bthc", wgt, xd[:, :, :, 2]).reshape(B, T, heads * ch)

    try:
        set_precision("f16x3")
        try:
            range_check("stale")
        except CdaeRangeError:
            pass
        g.manual_seed(22)
        y, exact = run()
        if which == "conv_weight":
            # weights are scale invariant since round 4 (per-tensor power-of-two scale of the weight planes): 2^22-scaled weights
            # (|w| up to ~5e5, far beyond the f16 maximum) compute CORRECTLY in the split mode, no flag
            range_check(which)
            assert torch.isfinite(y).all() and _rel(y, exact) < 6e-6, _rel(y, exact)
        else:
            with pytest.raises(CdaeRangeError):
                range_check(which)
        set_precision("fp32")
        g.manual_seed(22)
        y, exact = run()
        causaldiffae_amd.range_check(which + " in fp32 mode")
        assert torch.isfinite(y).all() and _rel(y, exact) < 1e-5
    finally:
        set_precision(prev)


@pytest.mark.gpu
def test_lds_out_of_range_reads_return_zero(tmp_path):
    """convwin.hip sends the fragment reads of padding taps to an LDS address beyond every allocation and relies on the hardware
    returning zeros there (with two blocks resident per CU, whose allocations are neighbours).  The probe reads such addresses from
    1024 blocks that filled their own 80 KB with a pattern: everything out of range must be zero, the in-range control must not."""
    import os, subprocess
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "hiptests", "lds_oob.hip")
    exe = str(tmp_path / "lds_oob")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", src, "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("offset")]
    assert len(lines) == 8
    for l in lines:
        nonzero = int(l.split(":")[1].split("of")[0])
        if "in range" in l:
            assert nonzero == 4096, l
        else:
            assert nonzero == 0, l


@pytest.mark.gpu
def test_library_lds_probe_confirms_zero_reads():
    """The check the library runs itself before the first window-conv launch on a device (cdae_convwin_lds_probe): 0 on this part;
    2 while the stream is capturing (nothing can be waited for there), and the capture stays valid."""
    from causaldiffae_amd._lib import lib, stream
    assert lib.cdae_convwin_lds_probe(stream()) == 0
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    x = torch.zeros(8, device="cuda:0")
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            rc = lib.cdae_convwin_lds_probe(stream())
            x += 1
    assert rc in (0, 2)            # (0: the per-device answer is cached after the first check)
    g.replay()
    torch.cuda.synchronize()
    assert x.sum().item() == 8


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,S,up", [(64, 128, 128, 64, False),      # 1024 tiles, four per persistent block
                                             (128, 256, 256, 32, False),     # K = 2304, two n-tiles share a window
                                             (128, 512, 512, 8, False),      # 8x8 level: split-K slabs + reduce
                                             (96, 384, 384, 16, False),      # 16x16: a tile spans exactly one image
                                             (256, 64, 128, 32, True)])      # sub-pixel phases (four 2x2-tap launches)
def test_window_conv_is_race_free_and_deterministic(N, Cin, Cout, S, up):
    """The second-generation window kernel keeps LDS-DMAs in flight across barriers with counted waits: a misplaced wait shows up as
    rare wrong tiles that come and go with memory load.  Twenty launches of the same problem, with an unrelated memory-bound kernel
    in between, must give bit-identical results, and the first must agree with an fp64 convolution."""
    from causaldiffae_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(31)
    x = ops.to_nhwc(torch.randn(N, Cin, S, S, device="cuda:0", generator=g))
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda:0", generator=g) / (9 * Cin) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device="cuda:0", generator=g)
    xs = _split_nhwc(x)
    junk = torch.empty(64 << 20, dtype=torch.float32, device="cuda:0")
    with torch.no_grad():
        first = ops.conv3x3_ps(xs, w, b, up=up).clone()
        for it in range(20):
            junk.normal_() if it % 3 == 0 else junk.mul_(1.0001)          # perturb caches / HBM queues between launches
            again = ops.conv3x3_ps(xs, w, b, up=up)
            assert torch.equal(again, first), it
        xin = F.interpolate(x[:2].contiguous(), scale_factor=2, mode="nearest") if up else x[:2].contiguous()
        exact = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
    assert (first[:2].double() - exact).abs().max().item() < 2e-5 * max(1.0, exact.abs().max().item())


ODD_WINDOW_SHAPES = [(5, 64, 96, 8, False),        # M = 320: second tile has 64 rows; N = 96 < 128 columns
                     (129, 512, 512, 8, True),     # odd batch at the 8 x 8 level: M % 256 = 64, four n-tiles, residual
                     (3, 128, 160, 16, False),     # M = 768 = 3 tiles of one image each; second n-tile 32 columns wide
                     (7, 96, 128, 32, True),       # Cin = 96: three 32-channel chunks
                     (1, 32, 32, 64, False),       # one image, one chunk (18 K-steps), 16 tiles
                     (2, 256, 384, 32, False)]     # three n-tiles


@pytest.mark.gpu
@pytest.mark.parametrize("splitk", [0, 1])
def test_window_conv_odd_shapes(splitk, expect_kernels):
    """Partial last tiles in M and N, Cin of three chunks, one-image and odd-batch problems on the second-generation window kernel,
    forced onto them by the dispatch threshold (cdae_tune_set; at production sizes the dispatcher routes e.g. batch 129 at the 8 x 8
    level there), with and without its K split; the launch log proves the window kernel produced every result."""
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import check, lib, ptr, stream, tune_scope, range_check
    g = torch.Generator(device="cuda:0").manual_seed(41)
    for (N, ci, co, S, res) in ODD_WINDOW_SHAPES:
        x = ops.to_nhwc(torch.randn(N, ci, S, S, device="cuda:0", generator=g))
        w = (torch.randn(co, ci, 3, 3, device="cuda:0", generator=g) / (9 * ci) ** 0.5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device="cuda:0", generator=g)
        r = ops.to_nhwc(torch.randn(N, co, S, S, device="cuda:0", generator=g)) if res else None
        planes = torch.empty((2, N, S, S, ci), dtype=torch.float16, device="cuda:0")
        check(lib.cdae_split_f16(ptr(x), ptr(planes[0]), ptr(planes[1]), x.numel(), stream()))
        with torch.no_grad(), tune_scope(convwin_min_tiles=1, convwin_splitk=splitk), expect_kernels(convwin=1):
            y = ops.conv3x3_ps(ops.SplitAct(planes[0], planes[1], (N, ci, S, S)), w, b, res=r)
        exact = F.conv2d(x.double().contiguous(), w.double(), b.double(), padding=1)
        if res:
            exact = exact + r.double()
        assert torch.isfinite(y).all()
        e = (y.double() - exact).abs().max().item() / max(1.0, exact.abs().max().item())
        assert e < 2e-5, ((N, ci, co, S, res), e)
    range_check("odd shapes")


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,C", [(3, 16, 16, 256), (2, 5, 7, 96), (2, 4, 6, 6), (1, 3, 3, 5)])
def test_sumpool2_matches_avg_pool(N, H, W, C):
    """cdae_sumpool2 (the gradient of the fused nearest-2x upsample: NHWC [N, 2H, 2W, C] -> [N, H, W, C] sums of 2 x 2 pixels): the four-
    channel vector form (C % 4 == 0) and the scalar form (C = 6, 5) against 4 * avg_pool2d; same order of additions, so exact."""
    import torch.nn.functional as F
    from causaldiffae_amd._lib import check, lib, ptr, stream
    g = torch.Generator(device="cuda:0").manual_seed(9)
    src = torch.randn(N, 2 * H, 2 * W, C, device="cuda:0", generator=g)
    dst = torch.empty(N, H, W, C, device="cuda:0")
    check(lib.cdae_sumpool2(ptr(src), ptr(dst), N, H, W, C, stream()))
    s = src.reshape(N, H, 2, W, 2, C)
    want = (s[:, :, 0, :, 0] + s[:, :, 0, :, 1]) + (s[:, :, 1, :, 0] + s[:, :, 1, :, 1])
    assert torch.equal(dst, want)
    ref = 4 * F.avg_pool2d(src.permute(0, 3, 1, 2).double(), 2).permute(0, 2, 3, 1)
    assert (dst.double() - ref).abs().max().item() < 1e-5


@pytest.mark.gpu
def test_conv_bank_follows_in_place_weight_change():
    """A banked conv3x3 weight changed IN PLACE by a torch op (autograd version moves, the weight epoch does not): the bank must refresh
    the weight's scale record BEFORE it rebuilds the planes.  Before round 5 `ConvWeightBank._refresh` asked the table for a refresh
    without naming the tensor, got none, built planes of w_new * 2^k_old and the kernel unscaled them with 2^-k_new: a silent factor
    2^(k_old - k_new) on the output (and f16 overflow one bit later).  Also: records rewritten on behalf of ANOTHER tensor make the
    bank rebuild (generation counter)."""
    import torch.nn.functional as F
    from causaldiffae_amd import ops
    from causaldiffae_amd.train_util import FlatParams
    dev = "cuda:0"
    g = torch.Generator().manual_seed(11)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Conv2d(32, 64, 3, padding=1).to(memory_format=torch.channels_last)
            self.b = torch.nn.Conv2d(64, 32, 3, padding=1).to(memory_format=torch.channels_last)

    m = M()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    m.to(dev)
    flat = FlatParams(m)
    assert flat.conv_bank is not None and flat.scale_table is not None
    x = torch.randn(2, 32, 16, 16, generator=g).to(dev)

    def run(conv, xin):
        with torch.no_grad():
            xs = ops.to_nhwc(xin)
            planes = torch.empty((2,) + tuple(xs.permute(0, 2, 3, 1).shape), dtype=torch.float16, device=dev)
            from causaldiffae_amd._lib import check, lib, ptr, ptr2, stream
            check(lib.cdae_split_f16(ptr(xs), *ptr2(planes), xs.numel(), stream()))
            out = ops.conv3x3_ps(ops.SplitAct(planes[0], planes[1], xs.shape), conv.weight, conv.bias)
        want = F.conv2d(xin.double().cpu(), conv.weight.detach().double().cpu(), conv.bias.detach().double().cpu(), padding=1)
        return (out.cpu().double() - want).abs().max().item() / want.abs().max().item()

    assert run(m.a, x) < 1e-5
    with torch.no_grad():
        m.a.weight.mul_(16.0)                       # version-only change: k moves by 4
    assert run(m.a, x) < 1e-5
    with torch.no_grad():
        m.a.weight.mul_(1.0 / 1024.0)
    assert run(m.a, x) < 1e-5
    # another tensor's refresh rewrites the shared records; the bank's planes of `a` must stay consistent with them
    y = torch.randn(2, 64, 16, 16, generator=g).to(dev)
    with torch.no_grad():
        m.b.weight.mul_(64.0)
    ops.weight_scale(m.b.weight)                    # refreshes the table on behalf of b (bank not involved)
    assert run(m.a, x) < 1e-5 and run(m.b, y) < 1e-5


@pytest.mark.gpu
def test_adamw_multi_ema_and_grad_scale():
    """cdae_adamw_ema_multi: AdamW on grad_scale * g + up to four EMA buffers in one pass == torch.optim.AdamW on the scaled gradient
    followed by the reference's update_ema per rate (train_util.py:292-297, nn.py:503-513)."""
    import ctypes
    from causaldiffae_amd._lib import check, lib, ptr, stream
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    n = 100003
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3.0
    rates = [0.999, 0.9999, 0.5]
    ref = torch.nn.Parameter(p0.double().clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    emas_ref = [p0.double().clone() for _ in rates]
    p, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    emas = [p0.to(dev).clone() for _ in rates]
    grad = gr.to(dev)
    ptrs = (ctypes.c_void_p * 3)(*[e.data_ptr() for e in emas])
    rs = (ctypes.c_double * 3)(*rates)
    for step in range(1, 4):
        ref.grad = gr.double() * 0.125
        opt.step()
        for e, r in zip(emas_ref, rates):
            e.mul_(r).add_(ref.detach(), alpha=1 - r)
        check(lib.cdae_adamw_ema_multi(ptr(p), ptr(grad), ptr(m), ptr(v), ptrs, rs, 3, n, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, 0.125, stream()))
    assert (p.cpu().double() - ref.detach()).abs().max().item() < 2e-6
    for e, er in zip(emas, emas_ref):
        assert (e.cpu().double() - er).abs().max().item() < 2e-6
    assert lib.cdae_adamw_ema_multi(ptr(p), ptr(grad), ptr(m), ptr(v), ptrs, rs, 5, n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, stream()) != 0


@pytest.mark.gpu
def test_box_calibration_is_plausible():
    """cdae_calib_mfma / cdae_calib_copy (bench.py's `box` record): a sustained f16 MFMA rate between half and all of the 2.5 PFLOP/s dense
    peak at a shader clock between 1 and 2.6 GHz, and an HBM copy rate between 2 and 8 TB/s."""
    from causaldiffae_amd import _lib
    box = _lib.calibrate(torch.device("cuda:0"), copy_bytes=1 << 29)
    assert 1200.0 < box["mfma_sustained_tflops"] < 2600.0, box
    assert 1.0 < box["sclk_under_load_ghz"] < 2.6, box
    assert 2.0 < box["hbm_copy_tbps"] < 8.5, box
