"""The 16-bit torso: autograd nodes of the training path whose activations and gradients are bf16 NHWC tensors.

Reference placement (improved_diffusion/unet.py:501-507 `convert_to_fp16`, fp16_util.py:9-15, nn.py:435-437): the input / middle /
output blocks carry half-precision activations end to end, GroupNorm32 computes in fp32 and casts back, master weights, embeddings,
softmax and the optimizer stay fp32.  Here the torso dtype is bf16 (BASELINE config [1] names it; gradients need no loss scaling):

  * a bf16 NHWC tensor IS the one-plane operand of the matrix-core kernels — no split pass, no second copy as planes;
  * conv3x3 / 1x1 epilogues round to bf16 once and leave the next GroupNorm's partial sums of the ROUNDED values;
  * GroupNorm32 forward / backward read and write bf16 rows with fp32 statistics (norm.hip, templated on the storage type);
  * weight gradients accumulate in fp32 straight into the flat gradient buffer, like the fp32-storage path.
  * the 1x1 convs / linears (skip_connection, qkv, proj_out) and their weight gradients are HBM streams at these row counts: the library
    runs them on rows16.hip / wg16.hip behind cdae_gemm16_ps / cdae_linear_wgrad_io (weight fragments in registers, rows through an
    LDS-DMA ring), smaller or odd shapes on the plane GEMM.

A ResBlock, an AttentionBlock and the Upsample conv are one autograd node each.  Shapes the 16-bit kernels do not take (the 4 x 4
level: rows shorter than the window kernels' minimum) run the fp32-storage nodes of ops.py between two casts.
"""
import weakref

import torch
from torch.autograd import Function

from . import ops
from ._lib import check, lib, ptr, ptr2, stream, workspace, workspace_bytes, WS_GN_PARTS

BF16 = torch.bfloat16


# ----------------------------------------------------------------------------- layout helpers
def new_act16(N, C, H, W, device):
    """Fresh bf16 activation: logical [N, C, H, W], NHWC storage."""
    return torch.empty_strided((N, C, H, W), (H * W * C, 1, W * C, C), dtype=BF16, device=device)


def rows16(x):
    """x as a dense bf16 NHWC-stored tensor (no copy when it already is)."""
    if x.dtype != BF16:
        return to16(x)
    if ops.is_nhwc(x):
        return x
    return x.contiguous(memory_format=torch.channels_last)


class _To16(Function):
    """fp32 -> bf16 activation (NHWC storage); the gradient comes back as fp32"""

    @staticmethod
    def forward(ctx, x):
        x = ops.to_nhwc(x)
        N, C, H, W = x.shape
        out = new_act16(N, C, H, W, x.device)
        n = x.numel()
        if n % 4 == 0:
            check(lib.cdae_cast_f32_bf16(ptr(x), ptr(out), n, stream()))
        else:
            out.copy_(x)
        return out

    @staticmethod
    def backward(ctx, d):
        return to32_raw(rows16(d))


class _To32(Function):
    """bf16 -> fp32 activation (NHWC storage); the gradient goes back as bf16"""

    @staticmethod
    def forward(ctx, x):
        return to32_raw(rows16(x))

    @staticmethod
    def backward(ctx, d):
        return to16_raw(ops.to_nhwc(d))


def to32_raw(x):
    N, C, H, W = x.shape
    out = ops.new_act(N, C, H, W, x.device)
    n = x.numel()
    if n % 4 == 0:
        check(lib.cdae_cast_bf16_f32(ptr(x), ptr(out), n, stream()))
    else:
        out.copy_(x)
    return out


def to16_raw(x):
    N, C, H, W = x.shape
    out = new_act16(N, C, H, W, x.device)
    n = x.numel()
    if n % 4 == 0:
        check(lib.cdae_cast_f32_bf16(ptr(x), ptr(out), n, stream()))
    else:
        out.copy_(x)
    return out


def to16(x):
    return x if x.dtype == BF16 else _To16.apply(x)


def to32(x):
    return x if x.dtype != BF16 else _To32.apply(x)


def torso16_on(model_or_none=None):
    """The 16-bit torso applies to grad-mode forwards in the mixed16 precision mode."""
    from ._lib import get_precision
    return ops._TORSO16_ON and torch.is_grad_enabled() and get_precision() == "mixed16"


# ----------------------------------------------------------------------------- bf16 copies of the 1x1 / linear weights
_W16 = {}


def w16(w):
    """Device pointer of the bf16 copy [N][K] of a dense fp32 weight.  Weights that live in a FlatParams buffer are served from ONE
    bf16 image of the whole buffer, rewritten by one cast per weight version (ops.register_flat16); others are cast on their own."""
    w = ops._root(w)
    hit = ops.flat16_pointer(w)
    if hit is not None:
        return hit
    tag = (w.data_ptr(), w._version, ops._WEIGHT_EPOCH[0])
    c = _W16.get(id(w))
    if c is None or c[0]() is not w or c[1] != tag:          # (weak reference: a new tensor may reuse a freed one's id AND address)
        if len(_W16) > 1024:
            for k in [k for k, v in _W16.items() if v[0]() is None]:
                del _W16[k]
        c = _W16[id(w)] = (weakref.ref(w), tag, w.detach().to(BF16).contiguous())
    return c[2].data_ptr()


def _lw16_ok(x, dy, ldx, lddy):
    """operands the grouped weight-gradient launch takes on bf16 rows (else the member keeps its own launch)"""
    return (ops._LW_GROUP_ON and x.dtype == BF16 and dy.dtype == BF16 and ldx % 4 == 0 and lddy % 4 == 0 and x.data_ptr() % 8 == 0 and dy.data_ptr() % 8 == 0)


def wt16(w):
    """Device pointer of the bf16 W^T [K][N] of a dense weight [N][K] (the hi plane of ops.wt_planes)"""
    return ops.wt_planes(w).data_ptr()


def _sk(dev):
    return ops._sk(dev)


def _gemm16(a16, lda, b16_ptr, ldb, bias, res, c, ldc, parts, M, N, K, io, accumulate=0):
    ws, wsb = _sk(c.device)
    check(lib.cdae_gemm16_ps(ptr(a16), lda, b16_ptr, ldb, ptr(bias), ptr(res), ptr(c), ldc, ptr(parts), M, N, K, io, accumulate, ws, wsb, stream()))


# ----------------------------------------------------------------------------- GroupNorm32 on bf16 rows
def _gn_stats(x, x2, N, C, HW, groups, eps, st):
    """(mean, rstd) [2, N, G] of the bf16 tensor(s): from the producers' partial sums where they left them, else one statistics pass"""
    dev = x.device
    C1 = x.shape[1]
    stats = torch.empty((2, N, groups), dtype=torch.float32, device=dev)
    p1, p2 = ops._rb_parts(x, HW), ops._rb_parts(x2, HW)
    if p1 is not None and (x2 is None or p2 is not None):
        check(lib.cdae_gn_stats_from_parts(ptr(p1[0]), C1, p1[1], ptr(p2[0]) if p2 else None, C - C1, p2[1] if p2 else 1, N, HW, groups, eps,
                                           *ptr2(stats), ptr(workspace(dev, "gnparts", workspace_bytes(WS_GN_PARTS, N, C))), st))
    else:
        gws = workspace(dev, "gn", 4 * lib.cdae_gn_workspace_floats(N, C))
        check(lib.cdae_gn_stats16(ptr(x), C1, ptr(x2), 0 if x2 is None else C - C1, C1, N, HW, C, groups, eps, *ptr2(stats),
                                  None, None, None, 0, None, ptr(gws), st))
    return stats


def _gn_apply(x, x2, stats, gamma, beta, ss, silu, N, C, H, W, groups, st):
    C1 = x.shape[1]
    y = torch.empty((N, H, W, C), dtype=BF16, device=x.device)
    check(lib.cdae_gn_apply16(ptr(x), C1, ptr(x2), 0 if x2 is None else C - C1, C1, ptr(y), C, N, H * W, C, groups, *ptr2(stats),
                              ptr(gamma), ptr(beta), ptr(ss), 2 * C if ss is None else ss.stride(0), 1 if silu else 0, st))
    return y


def _gn_bwd(x, x2, dy, dx, dx2, stats, gamma, beta, ss, silu, sinks, ss_sink, N, C, HW, groups, acc_dx, dx_add, st):
    """returns (dgamma, dbeta, dss) — None where the gradient went straight into its flat-buffer sink"""
    dev = x.device
    C1 = x.shape[1]
    (gg, rg), (gb_, rb_) = sinks
    direct = gg is not None and gb_ is not None
    dgamma = gg if direct else torch.empty_like(gamma)
    dbeta = gb_ if direct else torch.empty_like(beta)
    sink = ss_sink if ss is not None else None
    dss = None if ss is None else (sink if sink is not None else torch.empty((N, 2 * C), dtype=torch.float32, device=dev))
    gws = workspace(dev, "gn", 4 * lib.cdae_gn_workspace_floats(N, C))
    check(lib.cdae_gn_bwd16(ptr(x), C1, ptr(x2), 0 if x2 is None else C - C1, C1, ptr(dy), C, ptr(dx), C1 if x2 is not None else C,
                            ptr(dx2), 0 if x2 is None else C - C1, N, HW, C, groups, *ptr2(stats), ptr(gamma), ptr(beta),
                            ptr(ss), 2 * C if ss is None else ss.stride(0), 1 if silu else 0, ptr(dgamma), ptr(dbeta), 1 if direct else 0,
                            ptr(dss), 2 * C if dss is None else dss.stride(0), 1 if acc_dx else 0, ptr(dx_add), C, ptr(gws), st))
    if direct:
        dgamma = dbeta = None
        ops._done(rg, rb_)
    return dgamma, dbeta, (None if sink is not None else dss)


# ----------------------------------------------------------------------------- conv3x3 on bf16 rows
def _w16_conv(w):
    """(forward OHWI, forward K-group-major, dgrad [Cin][9][Cout], dgrad K-group-major) bf16 plane pointers of a conv3x3 weight"""
    bank = ops._bank(w)
    if bank is not None and bank.kpack:          # (the bank's one-launch bf16 forward planes ride with its K-group-major planes: `kpack` path toggle)
        return bank.pointers16(w)
    return ops.conv_planes16(w)


def conv_ok(N, H, W, Cin, Cout):
    """a stride-1 conv3x3 the 16-bit nodes take: channel counts the plane kernels accept.  Rows of 8 .. 64 pixels run the window kernels
    (forward / dgrad / grouped wgrad); shorter rows (the 4 x 4 level) the plane GEMM's conv gather for forward and dgrad and the
    fp32-operand wgrad on casts of the two (tiny) operands."""
    return Cin % 32 == 0 and Cout % 32 == 0 and W >= 4 and H >= 4 and long_ok(N, H, W, max(Cin, Cout))


def long_ok(N, H, W, C):
    return N * H * W * C < (1 << 31)


def _win_ok(N, H, W, Cin, Cout):
    return lib.cdae_conv3x3_wgrad_win_supported(N, H, W, Cin, Cout) == 1


def _conv_fwd(a16, w, b, res16, N, H, W, Cin, Cout, st, want_parts=True):
    dev = a16.device
    out = new_act16(N, Cout, H, W, dev)
    M = N * H * W
    parts = torch.empty((M // 32, Cout, 2), dtype=torch.float32, device=dev) if want_parts and (H * W) % 32 == 0 and Cout % 4 == 0 else None
    f, fk, _, _ = _w16_conv(w)
    ws, wsb = _sk(dev)
    check(lib.cdae_conv3x3_fwd16(ptr(a16), H * W * Cin, W * Cin, Cin, f, fk, ptr(b), ptr(res16), ptr(out), Cout, ptr(parts),
                                 N, H, W, Cin, Cout, ws, wsb, st))
    if parts is not None:
        out._gnparts = parts
    return out


def _conv_bwd(a16, dy16, w, sinks, has_b, N, H, W, Cin, Cout, need_w, st):
    """wgrad (fp32, into the flat-gradient sinks where they exist, on the side stream) and dgrad (bf16) of one stride-1 conv"""
    dev = dy16.device
    ws, wsb = _sk(dev)
    (gw, rw), (gb, rb) = sinks
    dw = db = None
    if need_w:
        direct = gw is not None and w.stride() == gw.stride() and (not has_b or gb is not None)
        if direct:
            dw, db = gw, gb
        else:
            dw = torch.empty_like(w)
            db = torch.empty(Cout, dtype=torch.float32, device=dev) if has_b else None

        def wg(st_, ws_, wsb_, dw=dw, db=db):
            check(lib.cdae_conv3x3_wgrad_win(ptr(a16), ptr(a16), ptr(dy16), ptr(dy16), ptr(dw), ptr(db), N, H, W, Cin, Cout,
                                             1 if direct else 0, ws_, wsb_, st_))
        if not _win_ok(N, H, W, Cin, Cout) and ops._IM2COL16_ON and Cin % 128 == 0 and Cout % 128 == 0 and N * H * W * 9 * Cin < (1 << 31):
            # rows too short for the window kernel (4 x 4): the patch matrix [pixels][9 Cin] (bf16, a few MB) and the streaming weight-gradient
            # kernel of the 1 x 1 convs (wg16.hip) — dW is [Cout][9 Cin] = the OHWI weight itself
            cols = torch.empty((N * H * W, 9 * Cin), dtype=BF16, device=dev)
            check(lib.cdae_im2col3x3_16(ptr(a16), ptr(cols), N, H, W, Cin, st))

            def wgc(st_, ws_, wsb_, dw=dw, db=db):
                check(lib.cdae_linear_wgrad_io(ptr(cols), 9 * Cin, ptr(dy16), Cout, ptr(dw), 9 * Cin, ptr(db), N * H * W, Cout, 9 * Cin, 12,
                                               1 if direct else 0, ws_, wsb_, st_))
            if direct:
                ops.side_launch(dev, (cols, dy16), wgc)
                dw = db = None
                ops._done(rw, rb)
            else:
                wgc(st, ws, wsb)
        elif not _win_ok(N, H, W, Cin, Cout):
            # (other channel counts) the fp32-operand implicit GEMM on casts of the two small operands
            a32, d32 = to32_raw(a16.permute(0, 3, 1, 2)), to32_raw(dy16.permute(0, 3, 1, 2))

            def wg32(st_, ws_, wsb_, dw=dw, db=db):
                check(lib.cdae_conv3x3_wgrad(ptr(a32), H * W * Cin, W * Cin, Cin, 1, ptr(d32), Cout, ptr(dw), ptr(db), N, H, W, Cin, Cout, 1, 0,
                                             1 if direct else 0, ws_, wsb_, st_))
            if direct:
                ops.side_launch(dev, (a32, d32), wg32)
                dw = db = None
                ops._done(rw, rb)
            else:
                wg32(st, ws, wsb)
        elif direct:
            ops.wgrad_win(dev, a16, dy16, dw, db, N, H, W, Cin, Cout)          # into the level's group launch
            dw = db = None
            ops._done(rw, rb)
        else:
            wg(st, ws, wsb)
    dyn = torch.empty((N, H, W, Cin), dtype=BF16, device=dev)
    _, _, d, dk = _w16_conv(w)
    check(lib.cdae_conv3x3_dgrad16(ptr(dy16), d, dk, ptr(dyn), Cin, N, H, W, Cin, Cout, ws, wsb, st))
    return dyn, dw, db


# ----------------------------------------------------------------------------- ResBlock
class _ResBlock16(Function):
    """The ResBlock (reference unet.py:156-199) on bf16 rows, one autograd node:
        h = conv1(silu(GN1(x)));  out = conv2(silu(GN2(h) * (1 + scale) + shift)) + skip(x),  skip = identity or 1x1 conv;
    x may be the skip concatenation [x | x2] read in place (unet.py:628)."""

    @staticmethod
    def forward(ctx, x, ss, ss_sink, g1, b1, w1, c1b, g2, b2, w2, c2b, sw, sb, groups, eps, x2=None):
        x = rows16(x)
        N, C1, H, W = x.shape
        if x2 is not None:
            x2 = rows16(x2)
        C = C1 + (0 if x2 is None else x2.shape[1])
        Cout = w1.shape[0]
        st = stream()
        dev = x.device
        w1_in, w2_in, w1, w2 = w1, w2, ops.ohwi(w1), ops.ohwi(w2)
        M = N * H * W
        stats1 = _gn_stats(x, x2, N, C, H * W, groups, eps, st)
        a1 = _gn_apply(x, x2, stats1, g1, b1, None, True, N, C, H, W, groups, st)
        h = _conv_fwd(a1, w1, c1b, None, N, H, W, C, Cout, st)
        if ss is not None:
            assert ss.shape == (N, 2 * Cout) and ss.stride(1) == 1 and ss.dtype == torch.float32
        stats2 = _gn_stats(h, None, N, Cout, H * W, groups, eps, st)
        a2 = _gn_apply(h, None, stats2, g2, b2, ss, True, N, Cout, H, W, groups, st)
        if sw is None:
            assert x2 is None
            skip = x
        else:                               # 1x1 skip conv on the bf16 rows: one plane GEMM per source
            skip = new_act16(N, Cout, H, W, dev)
            wp = w16(sw)
            _gemm16(x, C1, wp, C, sb, None, skip, Cout, None, M, Cout, C1, 1)
            if x2 is not None:
                _gemm16(x2, C - C1, wp + 2 * C1, C, None, skip, skip, Cout, None, M, Cout, C - C1, 3)
        out = _conv_fwd(a2, w2, c2b, skip, N, H, W, Cout, Cout, st)
        ctx.save_for_backward(x, h, ss, stats1, stats2, a1, a2, g1, b1, w1, g2, b2, w2, sw, x2)
        ctx.cfg = (groups, c1b is not None, c2b is not None, sb is not None)
        ctx.sinks = (ops._sink(g1), ops._sink(b1), ops._sink(w1_in), ops._sink(c1b), ops._sink(g2), ops._sink(b2), ops._sink(w2_in), ops._sink(c2b),
                     ops._sink(sw), ops._sink(sb))
        ctx.ss_sink = ss_sink if ss is not None else None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h, ss, stats1, stats2, a1, a2, g1, b1, w1, g2, b2, w2, sw, x2 = ctx.saved_tensors
        groups, has_c1b, has_c2b, has_sb = ctx.cfg
        sg1, sb1, sw1, sc1b, sg2, sb2, sw2, sc2b, ssw, ssb = ctx.sinks
        N, C1, H, W = x.shape
        C = C1 + (0 if x2 is None else x2.shape[1])
        Cout = w1.shape[0]
        dev = x.device
        st = stream()
        dout = rows16(dout)
        need = ctx.needs_input_grad
        M, HW = N * H * W, H * W
        # ---- second half: conv2, GN2 (the gradient between the halves stays bf16: it is conv1's dy)
        dyn2, dw2, dc2b = _conv_bwd(a2, dout, w2, (sw2, sc2b), has_c2b, N, H, W, Cout, Cout, need[9], st)
        dh = torch.empty((N, H, W, Cout), dtype=BF16, device=dev)
        dg2, db2, dss = _gn_bwd(h, None, dyn2, dh, None, stats2, g2, b2, ss, True, (sg2, sb2), ctx.ss_sink, N, Cout, HW, groups, False, None, st)
        del dyn2
        # ---- first half: conv1, then GN1 with the residual / skip gradient folded in
        dyn1, dw1, dc1b = _conv_bwd(a1, dh, w1, (sw1, sc1b), has_c1b, N, H, W, C, Cout, need[5], st)
        dsw = dsb = dx2 = None
        if sw is None:
            dx = new_act16(N, C, H, W, dev)
            dg1, db1, _ = _gn_bwd(x, None, dyn1, dx, None, stats1, g1, b1, None, True, (sg1, sb1), None, N, C, HW, groups, False, dout, st)
        else:
            (gsw, rsw), (gsb, rsb) = ssw, ssb
            direct = gsw is not None and gsw.is_contiguous() and (not has_sb or gsb is not None)
            dsw = gsw if direct else torch.empty_like(sw)
            dsb = (gsb if direct else torch.empty(Cout, dtype=torch.float32, device=dev)) if has_sb else None
            acc = 1 if direct else 0
            wt = wt16(sw)                     # bf16 W^T [C][Cout]
            dx = new_act16(N, C1, H, W, dev)
            _gemm16(dout, Cout, wt, Cout, None, None, dx, C1, None, M, C1, Cout, 1)
            if x2 is not None:
                C2 = C - C1
                dx2 = new_act16(N, C2, H, W, dev)
                _gemm16(dout, Cout, wt + 2 * C1 * Cout, Cout, None, None, dx2, C2, None, M, C2, Cout, 1)
            dg1, db1, _ = _gn_bwd(x, x2, dyn1, dx, dx2, stats1, g1, b1, None, True, (sg1, sb1), None, N, C, HW, groups, True, None, st)

            def wg(st_, ws_, wsb_, dsw=dsw, dsb=dsb):
                check(lib.cdae_linear_wgrad_io(ptr(x), C1, ptr(dout), Cout, ptr(dsw), C, ptr(dsb), M, Cout, C1, 12, acc, ws_, wsb_, st_))
                if x2 is not None:
                    check(lib.cdae_linear_wgrad_io(ptr(x2), C - C1, ptr(dout), Cout, dsw.data_ptr() + 4 * C1, C, None, M, Cout, C - C1, 12, acc, ws_, wsb_, st_))
            if direct and _lw16_ok(x, dout, C1, Cout) and (x2 is None or (_lw16_ok(x2, dout, C - C1, Cout) and C1 % 4 == 0)):
                ops.linear_wgrad(dev, x, C1, dout, Cout, ptr(dsw), C, ptr(dsb), M, Cout, C1, (x, dout), io=12)
                if x2 is not None:
                    ops.linear_wgrad(dev, x2, C - C1, dout, Cout, dsw.data_ptr() + 4 * C1, C, None, M, Cout, C - C1, (x2, dout), io=12)
                dsw = dsb = None
                ops._done(rsw, rsb if has_sb else None)
            elif direct:
                ops.side_launch(dev, (x, x2, dout), wg)
                dsw = dsb = None
                ops._done(rsw, rsb if has_sb else None)
            else:
                ws, wsb = _sk(dev)
                wg(st, ws, wsb)
        return dx, dss, None, dg1, db1, dw1, dc1b, dg2, db2, dw2, dc2b, dsw, dsb, None, None, dx2


def resblock_ok(x, Cout, groups=32):
    """x: bf16 tensor or CatAct of two"""
    if isinstance(x, ops.CatAct):
        N, C, H, W = x.shape
        c_ok = x.a.shape[1] % 32 == 0 and x.b.shape[1] % 32 == 0
    else:
        N, C, H, W = x.shape
        c_ok = True
    return (c_ok and C % groups == 0 and (C // groups) % 4 == 0 and (Cout // groups) % 4 == 0 and C % 32 == 0 and Cout % 32 == 0
            and conv_ok(N, H, W, C, Cout) and conv_ok(N, H, W, Cout, Cout))


def resblock_train(x, ss, g1, b1, w1, c1b, g2, b2, w2, c2b, sw=None, sb=None, groups=32, eps=1e-5):
    x2 = None
    if isinstance(x, ops.CatAct):
        x, x2 = x.a, x.b
    sink = getattr(ss, "_dss_sink", None) if ss is not None else None
    if ss is not None and ss.stride(-1) != 1:
        ss, sink = ss.contiguous(), None
    if sw is not None and sw.dim() != 2:
        w2d = sw.reshape(sw.shape[0], -1)
        gv = getattr(sw, "_grad_view", None)
        if gv is not None:
            w2d._grad_view, w2d._grad_ready = gv.reshape(sw.shape[0], -1), getattr(sw, "_grad_ready", None)
        sw = w2d
    out = _ResBlock16.apply(x, ss, sink, g1, b1, w1, c1b, g2, b2, w2, c2b, sw, sb, groups, eps, x2)
    return out


# ----------------------------------------------------------------------------- Upsample conv
class _UpConv16(Function):
    """out = conv3x3(nearest_2x(x), w) + b (reference unet.py:86-104) on bf16 rows"""

    @staticmethod
    def forward(ctx, x, w, b):
        x = rows16(x)
        N, C, H, W = x.shape
        Cout = w.shape[0]
        w_in, w = w, ops.ohwi(w)
        st = stream()
        up = torch.empty((N, 2 * H, 2 * W, C), dtype=BF16, device=x.device)
        check(lib.cdae_upsample2_16(ptr(x), ptr(up), N, H, W, C, st))
        out = _conv_fwd(up, w, b, None, N, 2 * H, 2 * W, C, Cout, st)
        ctx.save_for_backward(up, w)
        ctx.cfg = (b is not None, (N, C, H, W))
        ctx.sinks = (ops._sink(w_in), ops._sink(b))
        return out

    @staticmethod
    def backward(ctx, dy):
        up, w = ctx.saved_tensors
        has_b, (N, C, H, W) = ctx.cfg
        Cout = w.shape[0]
        st = stream()
        dy = rows16(dy)
        dxu, dw, db = _conv_bwd(up, dy, w, ctx.sinks, has_b, N, 2 * H, 2 * W, C, Cout, ctx.needs_input_grad[1], st)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_act16(N, C, H, W, dy.device)
            check(lib.cdae_sumpool2_16(ptr(dxu), ptr(dx), N, H, W, C, st))
        return dx, dw, db


def upconv_ok(x, Cout):
    N, C, H, W = x.shape
    return C % 32 == 0 and Cout % 32 == 0 and conv_ok(N, 2 * H, 2 * W, C, Cout)


def upconv_train(x, w, b=None):
    return _UpConv16.apply(x, w, b)


# ----------------------------------------------------------------------------- Downsample conv
class _Down16(Function):
    """out = conv3x3(x, w, stride 2, padding 1) + b (reference unet.py:92-105) with bf16 rows in and out.  Forward: the plane GEMM's strided
    gather on the bf16 tensor itself (one launch; it was the fp32-operand conv between two casts).  Backward: the kernels of the fp32-storage
    node (ops._Conv3x3: sub-pixel-phase dgrad, implicit-GEMM wgrad) on fp32 casts of the two operands — three launches per model."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = rows16(x)
        N, C, H, W = x.shape
        Cout = w.shape[0]
        w_in, w = w, ops.ohwi(w)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = new_act16(N, Cout, Ho, Wo, x.device)
        f, _, _, _ = _w16_conv(w)
        ws, wsb = _sk(x.device)
        check(lib.cdae_conv3x3_s2_fwd16(ptr(x), H * W * C, W * C, C, f, ptr(b), ptr(out), Cout, N, H, W, C, Cout, ws, wsb, stream()))
        ctx.save_for_backward(x, w)
        ctx.cfg = (b is not None, (N, C, H, W))
        ctx.sinks = (ops._sink(w_in), ops._sink(b))
        return out

    @staticmethod
    def backward(ctx, dy):
        x16, w = ctx.saved_tensors
        has_b, (N, C, H, W) = ctx.cfg
        Cout = w.shape[0]
        dev = dy.device
        dy32, x32 = to32_raw(rows16(dy)), to32_raw(x16)
        ws, wsb = _sk(dev)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx32 = ops.new_act(N, C, H, W, dev)
            if not ops._s2_dgrad_ps(dy32, w, dx32, N, H, W, C, Cout, ws, wsb):
                check(lib.cdae_conv3x3_dgrad(ptr(dy32), Cout, ptr(w), ptr(dx32), C, N, H, W, C, Cout, 2, 0, 0, ws, wsb, stream()))
            dx = to16_raw(dx32)
        if ctx.needs_input_grad[1]:
            (gw, rw), (gb, rb) = ctx.sinks
            direct = gw is not None and w.stride() == gw.stride() and (not has_b or gb is not None)
            if direct:
                dw, db = gw, gb
            else:
                dw = torch.empty_like(w)
                db = torch.empty(Cout, dtype=torch.float32, device=dev) if has_b else None

            def wg(st_, ws_, wsb_, dw=dw, db=db):
                check(lib.cdae_conv3x3_wgrad(ptr(x32), x32.stride(0), x32.stride(2), x32.stride(3), x32.stride(1), ptr(dy32), Cout, ptr(dw), ptr(db),
                                             N, H, W, C, Cout, 2, 0, 1 if direct else 0, ws_, wsb_, st_))
            if direct:
                ops.side_launch(dev, (x32, dy32), wg)
                dw = db = None
                ops._done(rw, rb)
            else:
                wg(stream(), ws, wsb)
        return dx, dw, db


def down_ok(x, Cout):
    N, C, H, W = x.shape
    return ops._DOWN16_ON and C % 32 == 0 and Cout % 32 == 0 and H % 2 == 0 and W % 2 == 0 and long_ok(N, H, W, max(C, Cout))


def downsample_train(x, w, b=None):
    return _Down16.apply(x, w, b)


# ----------------------------------------------------------------------------- AttentionBlock
class _AttnBlock16(Function):
    """The whole AttentionBlock (reference unet.py:223-253) as one node on a bf16 residual stream:
        out = x + proj(attention(qkv(GroupNorm(x)))).
    Where attn16.hip has the shape (T in {64, 256}) everything between the norm and the projection is bf16 rows and the attention
    core keeps no [T, T] tensor (log-sum-exp per query, probabilities recomputed in the backward); else (the 4 x 4 level) qkv / the
    core / its output stay fp32 as on the fp32-storage path.  The softmax is fp32 in registers either way (unet.py:250-252)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, wqkv, bqkv, wproj, bproj, heads, groups, eps):
        x = rows16(x)
        N, C, H, W = x.shape
        T, M = H * W, N * H * W
        dev = x.device
        st = stream()
        ch = C // heads
        core16 = lib.cdae_attn16_supported(T, ch) == 1
        stats = _gn_stats(x, None, N, C, T, groups, eps, st)
        xn = _gn_apply(x, None, stats, gamma, beta, None, False, N, C, H, W, groups, st)
        wq, wp = wqkv.reshape(3 * C, C), wproj.reshape(C, C)
        out = new_act16(N, C, H, W, dev)
        ws, wsb = _sk(dev)
        if core16:
            qkv = torch.empty((N, T, 3 * C), dtype=BF16, device=dev)
            _gemm16(xn, C, w16(wq), C, bqkv, None, qkv, 3 * C, None, M, 3 * C, C, 1)
            a = torch.empty((N, T, C), dtype=BF16, device=dev)
            aux = torch.empty((N * heads, T), dtype=torch.float32, device=dev)            # log-sum-exp per query
            check(lib.cdae_attn16_fwd(ptr(qkv), ptr(a), ptr(aux), N, T, heads, ch, st))
            _gemm16(a, C, w16(wp), C, bproj, x, out, C, None, M, C, C, 3)
        else:
            qkv = torch.empty((N, T, 3 * C), dtype=torch.float32, device=dev)
            _gemm16(xn, C, w16(wq), C, bqkv, None, qkv, 3 * C, None, M, 3 * C, C, 0)
            a = torch.empty((N, T, C), dtype=torch.float32, device=dev)
            aux = torch.empty((N * heads, T, T), dtype=torch.float32, device=dev)         # probabilities
            if lib.cdae_qkv_attention_fused_supported(T, ch):
                check(lib.cdae_qkv_attention_fwd_fused_p(ptr(qkv), ptr(a), ptr(aux), N, T, heads, ch, st))
            else:
                check(lib.cdae_qkv_attention_fwd(ptr(qkv), ptr(a), ptr(aux), N, T, heads, ch, st))
            check(lib.cdae_linear_fwd_io(ptr(a), C, ptr(wp), C, ptr(ops.weight_scale(wp)), ptr(bproj), ptr(x), ptr(out), C, M, C, C, 3, ws, wsb, st))
        ctx.save_for_backward(x, stats, xn, qkv, aux, a, gamma, beta, wq, wp)
        ctx.cfg = (heads, groups, bqkv is not None, bproj is not None, tuple(wqkv.shape), tuple(wproj.shape), core16)
        ctx.sinks = (ops._sink(gamma), ops._sink(beta), ops._sink(wqkv), ops._sink(bqkv), ops._sink(wproj), ops._sink(bproj))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, stats, xn, qkv, aux, a, gamma, beta, wq, wp = ctx.saved_tensors
        heads, groups, has_bq, has_bp, shape_q, shape_p, core16 = ctx.cfg
        sg, sb_, swq, sbq, swp, sbp = ctx.sinks
        N, C, H, W = x.shape
        T, M = H * W, N * H * W
        dev = x.device
        st = stream()
        ch = C // heads
        dout = rows16(dout)
        ws, wsb = _sk(dev)

        def wgrad(x_t, ldx, dy_t, lddy, sinks, has_b, Nw, K, io):
            (gw, rw), (gb, rb) = sinks
            direct = gw is not None and gw.is_contiguous() and (not has_b or gb is not None)
            dw = gw.reshape(Nw, K) if direct else torch.empty((Nw, K), dtype=torch.float32, device=dev)
            db = (gb if direct else torch.empty(Nw, dtype=torch.float32, device=dev)) if has_b else None

            def wg(st_, ws_, wsb_, dw=dw, db=db):
                check(lib.cdae_linear_wgrad_io(ptr(x_t), ldx, ptr(dy_t), lddy, ptr(dw), K, ptr(db), M, Nw, K, io, 1 if direct else 0, ws_, wsb_, st_))
            if direct and io == 12 and _lw16_ok(x_t, dy_t, ldx, lddy) and K % 4 == 0 and Nw % 4 == 0:
                ops.linear_wgrad(dev, x_t, ldx, dy_t, lddy, ptr(dw), K, ptr(db), M, Nw, K, (x_t, dy_t), io=12)
                ops._done(rw, rb if has_b else None)
                return None, None
            if direct:
                ops.side_launch(dev, (x_t, dy_t), wg)
                ops._done(rw, rb if has_b else None)
                return None, None
            wg(st, ws, wsb)
            return dw, db

        dxn = torch.empty((N, H, W, C), dtype=BF16, device=dev)
        if core16:
            # ---- proj: da = dout @ Wp, dWp = dout^T a;  core: dqkv from q, k, v, out, lse;  qkv: dxn = dqkv @ Wq, dWq = dqkv^T xn — all bf16 rows
            da = torch.empty((N, T, C), dtype=BF16, device=dev)
            _gemm16(dout, C, wt16(wp), C, None, None, da, C, None, M, C, C, 1)
            dwp, dbp = wgrad(a, C, dout, C, (swp, sbp), has_bp, C, C, 12)
            dqkv = torch.empty_like(qkv)
            dsum = torch.empty_like(aux)
            check(lib.cdae_attn16_bwd(ptr(qkv), ptr(a), ptr(da), ptr(aux), ptr(dsum), ptr(dqkv), N, T, heads, ch, st))
            del da
            _gemm16(dqkv, 3 * C, wt16(wq), 3 * C, None, None, dxn, C, None, M, C, 3 * C, 1)
            dwq, dbq = wgrad(xn, C, dqkv, 3 * C, (swq, sbq), has_bq, 3 * C, C, 12)
        else:
            da = torch.empty((N, T, C), dtype=torch.float32, device=dev)
            _gemm16(dout, C, wt16(wp), C, None, None, da, C, None, M, C, C, 0)
            dwp, dbp = wgrad(a, C, dout, C, (swp, sbp), has_bp, C, C, 4)
            dqkv = torch.empty_like(qkv)
            dprobs = torch.empty_like(aux)
            check(lib.cdae_qkv_attention_bwd(ptr(qkv), ptr(aux), ptr(da), ptr(dqkv), ptr(dprobs), N, T, heads, ch, st))
            del dprobs, da
            check(lib.cdae_linear_dgrad_io(ptr(dqkv), 3 * C, ptr(wq), C, ptr(dxn), C, M, 3 * C, C, 1, ws, wsb, st))
            dwq, dbq = wgrad(xn, C, dqkv, 3 * C, (swq, sbq), has_bq, 3 * C, C, 8)
        # ---- GroupNorm backward with the residual gradient folded in
        dx = new_act16(N, C, H, W, dev)
        dg, db, _ = _gn_bwd(x, None, dxn, dx, None, stats, gamma, beta, None, False, (sg, sb_), None, N, C, T, groups, False, dout, st)
        return dx, dg, db, (None if dwq is None else dwq.reshape(shape_q)), dbq, (None if dwp is None else dwp.reshape(shape_p)), dbp, None, None, None


def attn_ok(x, heads):
    N, C, H, W = x.shape
    return C % 32 == 0 and (C // 32) % 4 == 0 and (H * W) % 4 == 0 and C % heads == 0


def attention_block(x, norm, qkv, proj, heads):
    return _AttnBlock16.apply(x, norm.weight, norm.bias, qkv.weight, qkv.bias, proj.weight, proj.bias, heads, norm.num_groups, norm.eps)
