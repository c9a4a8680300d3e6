"""UNetModel noise predictor with the reference's constructor, forward signature and state-dict layout
(reference improved_diffusion/unet.py), executed by the HIP kernels behind ops.py.

Fusion map (what one launch replaces in the reference's eager ATen stream; kernels and dispatch table: DESIGN.md §4-5):
  GroupNorm32 statistics                        -> the producing conv's epilogue partial sums + one small finish launch (with the
                                                   folded (a, b) table), or gn_partial + finalize where no producer left sums
  GroupNorm32 + (1+scale)*.+shift + SiLU        -> written ONCE as the next conv's f16 hi/lo operand planes, group-major
                                                   (gn_apply_gm; training: f16 + bf16 planes)                 (unet.py:186,190-194)
  ResBlock entry with a 1x1 skip conv           -> skipgn_kernel: skip GEMM + first GroupNorm's planes from one sweep over the
                                                   (concatenated, never materialised) block input              (unet.py:143-147,165-171)
  conv3x3 + bias + residual (+ next GN's sums)  -> convwin_kernel on the planes (window resident in LDS)      (unet.py:186,194,198)
  F.interpolate(nearest 2x) + conv3x3           -> four folded 2x2 sub-pixel convs of the LOW-res input, one launch (unet.py:76-78)
  GroupNorm -> qkv 1x1                          -> skipgn_kernel<NORMA>: the norm applied to the rows as they stream  (unet.py:226)
  QKVAttention                                  -> attn_fused_kernel: QK^T, fp32 softmax in registers, PV — one launch (unet.py:239-253)
  proj_out 1x1 + residual                       -> streaming GEMM over NHWC rows                              (unet.py:230-231)
  22 emb_layers Linear(SiLU(emb))               -> ONE batched GEMM per forward                               (unet.py:148-154)
  output head GN -> SiLU -> conv3x3 (C <= 8)    -> head_mfma_kernel, exact fp32 products on v_mfma_f32_4x4x1    (unet.py:474-478)
Training: a ResBlock is one autograd node (ops.resblock_train), the Upsample conv another, the embedding projections a third.
"""
from abc import abstractmethod

import torch as th
import torch.nn as nn

from . import ops, ops16
from .nn import (
    CausalModeling,
    ConvNd,
    Dropout,
    Embedding,
    GaussianConvEncoder,
    Identity,
    Linear,
    MultivariateCausalFlow,
    avg_pool_nd,
    SiLU,
    _RNG_OVERRIDE,
    checkpoint,
    conv_nd,
    linear,
    normalization,
    reparameterize,
    timestep_embedding,
    zero_module,
)

# Adjacency matrices (unet.py:571-578; pendulum is the commented-out alternative the test script passes itself)
ADJACENCY = {
    "morphomnist": [[0, 1], [0, 0]],
    "circuit": [[0, 1, 1, 1], [0, 0, 0, 1], [0, 0, 0, 1], [0, 0, 0, 0]],
    "pendulum": [[0, 0, 1, 1], [0, 0, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]],
}


def encoder_hidden_dims(image_size, n_vars):
    """Encoder depth that ends on a 2x2 map (fc in-features = dims[-1]*4, reference nn.py:57).  The reference
    hard-codes the 6-conv list (unet.py:377), which only runs at 96/128 px."""
    if image_size in (28, 32):
        return [16, 32, 64, 128]
    if image_size == 64:
        return [16, 32, 32, 64, 128]
    return [16, 32, 32, 64, 64, 128]


class TimestepBlock(nn.Module):
    @abstractmethod
    def forward(self, x, emb):
        """Apply the module to `x` given `emb` timestep embeddings."""


_ADJ_CACHE = {}


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def forward(self, x, emb):
        for layer in self:
            x = layer(x, emb) if isinstance(layer, TimestepBlock) else layer(x)
        return x


class EmbAll:
    """The timestep embedding together with every ResBlock's `emb_layers` projection of it, computed by ONE GEMM per forward
    (inference path): `slices[id(block)]` is that block's [N, (2)Cout] column slice of the batched result."""
    __slots__ = ("emb", "slices")

    def __init__(self, emb, slices):
        self.emb, self.slices = emb, slices


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2):
        super().__init__()
        if dims != 2:
            raise NotImplementedError("Upsample: 2-D only (the reference's scripts never build another)")
        self.channels, self.use_conv, self.dims = channels, use_conv, dims
        if use_conv:
            self.conv = conv_nd(dims, channels, channels, 3, padding=1)

    def forward(self, x):
        assert x.shape[1] == self.channels
        if not self.use_conv:                  # conv_resample=False (unet.py:76-78 without the conv): nearest 2x, nothing else
            return ops16.to16(ops.upsample2(ops16.to32(x))) if x.dtype == th.bfloat16 else ops.upsample2(x)
        if x.dtype == th.bfloat16:             # the 16-bit torso
            if ops16.upconv_ok(x, self.channels):
                return ops16.upconv_train(x, self.conv.weight, self.conv.bias)
            return ops16.to16(self.forward(ops16.to32(x)))
        xs = getattr(x, "_split", None)        # pre-split planes left by the producing kernel (inference): sub-pixel form
        if xs is not None and ops.presplit_ok():
            return self.conv(xs, up=True, gn_stats=True)
        if ops.upconv3x3_train_ok(x, self.channels):
            return ops.upconv3x3_train(x, self.conv.weight, self.conv.bias)
        return self.conv(x, up=True)           # nearest-2x folded into the conv's input gather


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2):
        super().__init__()
        if dims != 2:
            raise NotImplementedError("Downsample: 2-D only (the reference's scripts never build another)")
        self.channels, self.use_conv, self.dims = channels, use_conv, dims
        self.op = conv_nd(dims, channels, channels, 3, stride=2, padding=1) if use_conv else avg_pool_nd(dims, 2)

    def forward(self, x):
        assert x.shape[1] == self.channels
        if not self.use_conv:                  # conv_resample=False (unet.py:101-103): 2 x 2 average pool
            return ops16.to16(self.op(ops16.to32(x))) if x.dtype == th.bfloat16 else self.op(x)
        if x.dtype == th.bfloat16:             # the 16-bit torso: forward on the bf16 rows themselves, backward on the fp32-storage node's kernels
            if ops16.down_ok(x, self.channels):
                return ops16.downsample_train(x, self.op.weight, self.op.bias)
            return ops16.to16(self.op(ops16.to32(x)))
        xs = getattr(x, "_split", None)
        if xs is not None and ops.presplit_ok():
            return self.op(xs, gn_stats=True)
        return self.op(x)


class ResBlock(TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False):
        super().__init__()
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_conv, self.use_checkpoint, self.use_scale_shift_norm = use_conv, use_checkpoint, use_scale_shift_norm
        self.emit_split = False        # set by UNetModel when a Down/Upsample conv reads this block's output
        self.in_layers = nn.Sequential(normalization(channels), SiLU(), conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(
            SiLU(), linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), SiLU(), Dropout(p=dropout),
            zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def forward(self, x, emb):
        return checkpoint(self._forward, (x, emb), self.parameters(), self.use_checkpoint)

    def _train_fused(self, x):
        """Grad-mode fast path: both GroupNorm -> SiLU -> conv3x3 chains as fused nodes on the pre-split kernels."""
        n1, c1, n2, c2 = self.in_layers[0], self.in_layers[2], self.out_layers[0], self.out_layers[3]
        if isinstance(x, ops.CatAct):      # the concatenation is read in place only by the whole-block node with a 1x1 skip conv
            if not (self.use_scale_shift_norm and ops.resblock_node_ok() and isinstance(self.skip_connection, ConvNd) and self.skip_connection.kernel_size == 1):
                return False
        return ((self.dropout == 0 or not self.training) and len(x.shape) == 4
                and ops.train_presplit_ok(x, c1.out_channels, n1.num_groups)
                and ops.train_presplit_ok((x.shape[0], c1.out_channels, x.shape[2], x.shape[3]), c2.out_channels, n2.num_groups))

    def _forward_train(self, x, emb):
        n1, c1, n2, c2 = self.in_layers[0], self.in_layers[2], self.out_layers[0], self.out_layers[3]
        if not isinstance(x, ops.CatAct):
            x = ops.to_nhwc(x)
        emb_out = emb.slices[id(self)] if isinstance(emb, EmbAll) else self.emb_layers[1](ops.silu(emb))
        sk = self.skip_connection
        if self.use_scale_shift_norm and ops.resblock_node_ok() and (isinstance(sk, Identity) or sk.kernel_size == 1):
            one = isinstance(sk, Identity)           # the whole block as one autograd node
            return ops.resblock_train(x, emb_out, n1.weight, n1.bias, c1.weight, c1.bias, n2.weight, n2.bias, c2.weight, c2.bias,
                                      None if one else sk.weight, None if one else sk.bias, n1.num_groups, n1.eps)
        h = ops.gn_conv3x3(x, n1.weight, n1.bias, None, c1.weight, c1.bias, None, True, n1.num_groups, n1.eps)
        skip = x if isinstance(self.skip_connection, Identity) else self.skip_connection(x)
        if self.use_scale_shift_norm:
            return ops.gn_conv3x3(h, n2.weight, n2.bias, emb_out, c2.weight, c2.bias, skip, True, n2.num_groups, n2.eps)
        return ops.gn_conv3x3(h + emb_out[:, :, None, None], n2.weight, n2.bias, None, c2.weight, c2.bias, skip, True, n2.num_groups, n2.eps)

    def _forward16(self, x, emb):
        """bf16 residual stream (ops16): the whole block as one node where the 16-bit kernels take the shape, else the fp32-storage
        path between two casts (the 4 x 4 level)"""
        n1, c1, n2, c2 = self.in_layers[0], self.in_layers[2], self.out_layers[0], self.out_layers[3]
        sk = self.skip_connection
        cat = isinstance(x, ops.CatAct)
        if (self.use_scale_shift_norm and (self.dropout == 0 or not self.training) and (isinstance(sk, Identity) or sk.kernel_size == 1)
                and not (cat and isinstance(sk, Identity)) and ops16.resblock_ok(x, c1.out_channels, n1.num_groups)):
            emb_out = emb.slices[id(self)] if isinstance(emb, EmbAll) else self.emb_layers[1](ops.silu(emb))
            one = isinstance(sk, Identity)
            return ops16.resblock_train(x, emb_out, n1.weight, n1.bias, c1.weight, c1.bias, n2.weight, n2.bias, c2.weight, c2.bias,
                                        None if one else sk.weight, None if one else sk.bias, n1.num_groups, n1.eps)
        x32 = ops16.to32(th.cat([x.a, x.b], dim=1).contiguous(memory_format=th.channels_last)) if cat else ops16.to32(x)
        return ops16.to16(self._forward(x32, emb))

    def _forward(self, x, emb):
        if (x.a if isinstance(x, ops.CatAct) else x).dtype == th.bfloat16:
            return self._forward16(x, emb)
        if th.is_grad_enabled():
            if self._train_fused(x):
                return self._forward_train(x, emb)
            if isinstance(x, ops.CatAct):
                x = ops.materialize(x)                              # grad mode off the fused path: a real (differentiable) concatenation
        if not isinstance(x, ops.CatAct):                          # CatAct: the skip concatenation, read in place by GN and the 1x1 skip
            x = ops.to_nhwc(x)
        sk = self.skip_connection
        # (coef: a 1x1 skip conv will apply this GroupNorm itself while it streams x — the statistics launch then writes its (a, b) table too)
        h = self.in_layers[0](x, silu=True, split=True, coef=isinstance(sk, ConvNd) and sk.kernel_size == 1)      # GN + SiLU (pre-split f16 planes on the inference path)
        skip = None
        if isinstance(h, ops.LazyGN) and isinstance(sk, ConvNd) and sk.kernel_size == 1 and ops.skip_gn_ok(h, sk.weight):
            N_, C_, H_, W_ = h.shape
            # the 1x1 skip conv writes the normalised planes (group-major where the next conv runs on the window kernel) while it reads x
            skip, h = ops.skip_gn_fused(h, sk.weight, sk.bias, gm=ops.gm_wanted(N_, H_, W_, C_, self.in_layers[2].weight.shape[0]))
        fast = isinstance(h, (ops.SplitAct, ops.LazyGN))
        h = self.in_layers[2](h, gn_stats=True) if fast else self.in_layers[2](h)     # conv3x3 + bias (+ GroupNorm partial sums)
        if isinstance(emb, EmbAll):
            emb_out = emb.slices[id(self)]                         # column slice of the batched emb_layers GEMM
        else:
            emb_out = self.emb_layers[1](ops.silu(emb))            # [N, (2)Cout]
        split = not (self.dropout > 0 and self.training)           # an active dropout needs the normalised values as a plain tensor
        if self.use_scale_shift_norm:
            h = self.out_layers[0](h, scale_shift=emb_out, silu=True, split=split)
        else:
            h = self.out_layers[0](h + emb_out[:, :, None, None], silu=True, split=split)
        h = self.out_layers[2](h)                                  # nn.Dropout: identity unless training with p > 0
        if skip is None:
            skip = ops.materialize(x) if isinstance(self.skip_connection, Identity) else self.skip_connection(x)
        if isinstance(h, (ops.SplitAct, ops.LazyGN)):              # emit_split: a Down/Upsample conv consumes this block's output
            return self.out_layers[3](h, res=skip, emit_split=self.emit_split, gn_stats=True)
        return self.out_layers[3](h, res=skip)                     # conv3x3 + bias + residual


class QKVAttention(nn.Module):
    """[N*H, 3*ch, T] -> [N*H, ch, T] like the reference module (unet.py:239-253); the fused block below
    uses the row-major form directly."""

    def forward(self, qkv):
        BH, C3, T = qkv.shape
        rows = qkv.permute(0, 2, 1).contiguous()                   # [BH, T, 3ch], heads=1 per batch entry
        out = ops.qkv_attention(rows, 1)
        return out.permute(0, 2, 1)


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads=1, use_checkpoint=False):
        super().__init__()
        self.channels, self.num_heads, self.use_checkpoint = channels, num_heads, use_checkpoint
        self.emit_split = False
        self.norm = normalization(channels)
        self.qkv = conv_nd(1, channels, channels * 3, 1)
        self.attention = QKVAttention()
        self.proj_out = zero_module(conv_nd(1, channels, channels, 1))

    def forward(self, x):
        return checkpoint(self._forward, (x,), self.parameters(), self.use_checkpoint)

    def _forward(self, x):
        if x.dtype == th.bfloat16:             # the 16-bit torso: the block as one node on the bf16 residual stream
            if ops16.attn_ok(x, self.num_heads):
                return ops16.attention_block(x, self.norm, self.qkv, self.proj_out, self.num_heads)
            return ops16.to16(self._forward(ops16.to32(x)))
        x = ops.to_nhwc(x)
        N, C, H, W = x.shape
        T = H * W
        h = self.norm(x, split=True, coef=True)                                     # GN, no activation (folded into the qkv GEMM where that is built)
        qkv = None
        if isinstance(h, ops.LazyGN):
            w2 = self.qkv.weight.reshape(self.qkv.weight.shape[0], -1)
            if ops.linear_gn_ok(h, w2):
                qkv = ops.linear_gn(h, w2, self.qkv.bias)                           # norm -> qkv in one pass over x (no normalised tensor)
            else:
                h = h.planes()
        if qkv is not None:
            pass
        elif isinstance(h, ops.SplitAct):
            qkv = ops.linear_ps(h, self.qkv.weight, self.qkv.bias)
        else:
            rows = h.permute(0, 2, 3, 1).reshape(N * T, C)
            qkv = ops.linear(rows, self.qkv.weight, self.qkv.bias)                  # [N*T, 3C]; channel = head*3ch + {q,k,v}*ch + d
        a = ops.qkv_attention(qkv.reshape(N, T, 3 * C), self.num_heads)            # [N, T, C]
        xr = x.permute(0, 2, 3, 1).reshape(N * T, C)
        if self.emit_split and ops.presplit_ok():
            out, planes = ops.linear_emit(a.reshape(N * T, C), self.proj_out.weight, self.proj_out.bias, xr, (N, C, H, W))
            out = out.reshape(N, H, W, C).permute(0, 3, 1, 2)
            out._split = planes
            return out
        w2 = self.proj_out.weight.reshape(C, C)
        hit = None if th.is_grad_enabled() else ops.linear_stream_gn(a.reshape(N * T, C), w2, self.proj_out.bias, xr, T)
        if hit is not None:                     # streaming GEMM whose epilogue leaves the next ResBlock's GroupNorm sums on the result
            out = hit[0].reshape(N, H, W, C).permute(0, 3, 1, 2)
            out._gnparts = hit[1]
            return out
        out = ops.linear(a.reshape(N * T, C), self.proj_out.weight, self.proj_out.bias, res=xr)
        return out.reshape(N, H, W, C).permute(0, 3, 1, 2)


class UNetModel(nn.Module):
    """The full UNet with timestep / label / context / representation conditioning (reference unet.py:279-632)."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, c_dim=None, rep_dim=None,
                 causal_modeling=False, flow_based=False, use_checkpoint=False, num_heads=1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, masking=False, n_vars=4, image_size=None, encoder_hidden=None):
        super().__init__()
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, attention_resolutions
        self.dropout, self.channel_mult, self.conv_resample = dropout, channel_mult, conv_resample
        self.num_classes, self.c_dim, self.rep_dim = num_classes, c_dim, rep_dim
        self.use_checkpoint, self.num_heads, self.num_heads_upsample = use_checkpoint, num_heads, num_heads_upsample
        self.causal_modeling, self.flow_based, self.masking = causal_modeling, flow_based, masking
        self.drop_prob = 0.5
        self.n_vars = n_vars
        self.adjacency = None            # optional override of the graph hard-coded in the reference forward (Q2)

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, time_embed_dim), SiLU(), linear(time_embed_dim, time_embed_dim))
        if num_classes is not None:
            self.label_emb = Embedding(num_classes, time_embed_dim)
        if c_dim is not None:
            self.c_emb = nn.Sequential(linear(c_dim, 256), SiLU(), linear(256, time_embed_dim))
        if rep_dim is not None:
            hidden = encoder_hidden or (encoder_hidden_dims(image_size, n_vars) if image_size else None)
            self.rep_emb = GaussianConvEncoder(in_channels=in_channels, latent_dim=rep_dim, hidden_dims=hidden, num_vars=4)
            self.up_emb = Linear(rep_dim, time_embed_dim)
        if causal_modeling:
            self.causal_mask = CausalModeling(latent_dim=rep_dim, num_var=n_vars, learn=False)
        if flow_based:        # the reference builds MultivariateCausalFlow(dim=2, k=256) whatever n_vars is (unet.py:385-386)
            self.causal_flow = MultivariateCausalFlow(dim=2, k=256)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        input_block_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, time_embed_dim, dropout, out_channels=mult * model_channels, dims=dims,
                                   use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, use_checkpoint=use_checkpoint, num_heads=num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims)))
                input_block_chans.append(ch)
                ds *= 2

        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm),
            AttentionBlock(ch, use_checkpoint=use_checkpoint, num_heads=num_heads),
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm),
        )

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [ResBlock(ch + input_block_chans.pop(), time_embed_dim, dropout, out_channels=model_channels * mult,
                                   dims=dims, use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, use_checkpoint=use_checkpoint, num_heads=num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))

        self.out = nn.Sequential(normalization(ch), SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))
        # producers whose output feeds a resampling conv also emit it as pre-split planes on the inference path
        for i, blk in enumerate(self.input_blocks):
            if isinstance(blk[0], Downsample) and isinstance(self.input_blocks[i - 1][-1], (ResBlock, AttentionBlock)):
                self.input_blocks[i - 1][-1].emit_split = True
        for blk in self.output_blocks:
            if isinstance(blk[-1], Upsample) and isinstance(blk[-2], (ResBlock, AttentionBlock)):
                blk[-2].emit_split = True

    # ------------------------------------------------------------------ precision
    def convert_to_fp16(self):
        """Reduced-precision torso (reference unet.py:501-507 casts the conv weights to half, fp16_util.py:9-15).  In grad mode (training)
        every activation and gradient from the input conv's result to the output head's input is then a bf16 NHWC tensor (ops16.py: the
        one-plane operand of the matrix-core kernels, fp32 accumulation, results rounded once in the epilogue); master weights, weight
        gradients, GroupNorm statistics (GroupNorm32, nn.py:435-437), the softmax, embeddings, the encoder and the optimizer stay fp32.  In
        no-grad mode storage stays fp32 and the contractions use one f16 plane per operand.  The mode belongs to this model: other models /
        samplers of the process keep the parity mode."""
        self._cdae_precision = "mixed16"       # applied around THIS model's forward (and by TrainLoop around its backward), not process-wide

    def convert_to_fp32(self):
        self._cdae_precision = None

    @property
    def inner_dtype(self):
        return next(self.input_blocks.parameters()).dtype

    # ------------------------------------------------------------------ conditioning
    def default_adjacency(self, device):
        """Graph used when the encoder path runs inside forward (reference unet.py:571-578): n_vars==2 ->
        MorphoMNIST, else CausalCircuit; `self.adjacency` overrides it (the reference ignores its A argument)."""
        A = self.adjacency
        if A is None:
            A = ADJACENCY["morphomnist"] if self.n_vars == 2 else ADJACENCY["circuit"]
        if isinstance(A, th.Tensor):
            return A.to(device=device, dtype=th.float32)
        key = (str(device), tuple(map(tuple, A)))          # uploaded once: no host-to-device copy inside a (graph-captured) step
        hit = _ADJ_CACHE.get(key)
        if hit is None:
            hit = _ADJ_CACHE[key] = th.as_tensor(A, dtype=th.float32).to(device)
        return hit

    def embed(self, timesteps, y=None, c=None, x_start=None, z=None):
        """Everything ahead of the conv torso: returns (emb, mu, var, z_post, mask)."""
        emb = self.time_embed[2](self.time_embed[0](timestep_embedding(timesteps, self.model_channels), act=ops.ACT_SILU))
        if self.num_classes is not None:
            assert y.shape == (timesteps.shape[0],)
            emb = ops.embedding_add(emb, self.label_emb.weight, y)
        if self.c_dim is not None:
            emb = self.c_emb[2](self.c_emb[0](c.float(), act=ops.ACT_SILU), res=emb)
        mu = var = z_post = mask = None
        if self.rep_dim is not None:
            if z is None:
                mu, var = self.rep_emb.encode(x_start)
                if self.causal_modeling:
                    A = self.default_adjacency(mu.device)
                    if self.flow_based:       # unet.py:580-587: flow in place of the masked MLPs; `mask` becomes a scalar
                        Cm = th.eye(A.shape[0], device=mu.device) - A
                        z_post, _ = self.causal_flow.flow(mu, Cm)
                        log_det, _ = self.causal_flow.reverse(z_post, Cm)
                        mask = -th.mean(log_det)
                    else:
                        z_pre = self.causal_mask.causal_masking(mu, A)
                        z_post = self.causal_mask.nonlinearity_add_back_noise(mu, z_pre)
                    z = reparameterize(z_post, var * 0.001)
                else:
                    z = reparameterize(mu, var * 0.001)
                if self.masking:
                    mask = _RNG_OVERRIDE.get("cfg_mask")
                    if mask is None:
                        mask = th.bernoulli(th.full((z.shape[0],), 1 - self.drop_prob, device=z.device))
                    mask = mask.to(z.device).float()
                    z = z * mask[:, None]
                    z_post = z_post * mask[:, None]
            emb = self.up_emb(z.float(), res=emb)
        return emb, mu, var, z_post, mask

    def emb_param_groups(self):
        """(blocks, weights, biases) of the ResBlocks' `emb_layers` Linear in module order: train_util.FlatParams stores them
        adjacently so that their concatenation is a view (ops._EmbAllTrain)."""
        blocks = [m for m in self.modules() if isinstance(m, ResBlock)]
        return blocks, [b.emb_layers[1].weight for b in blocks], [b.emb_layers[1].bias for b in blocks]

    def _emb_all(self, emb):
        """All 22 `emb_layers` projections (unet.py:186: Linear(SiLU(emb)) per ResBlock) as one GEMM over the concatenated
        weights; the concatenation is cached per weight version."""
        blocks = [m for m in self.modules() if isinstance(m, ResBlock)]
        tag = (ops._WEIGHT_EPOCH[0],) + tuple(b.emb_layers[1].weight._version for b in blocks) + tuple(b.emb_layers[1].weight.data_ptr() for b in blocks)
        cache = getattr(self, "_emb_cat", None)
        if cache is None or cache[0] != tag:
            w = th.cat([b.emb_layers[1].weight.detach() for b in blocks], dim=0).contiguous()
            bias = th.cat([b.emb_layers[1].bias.detach() for b in blocks], dim=0).contiguous()
            offs, o = [], 0
            for b in blocks:
                offs.append((id(b), o, b.emb_layers[1].weight.shape[0]))
                o += b.emb_layers[1].weight.shape[0]
            cache = (tag, w, bias, offs)
            self._emb_cat = cache
        _, w, bias, offs = cache
        e_all = ops.linear(ops.silu(emb), w, bias)
        return EmbAll(emb, {bid: e_all[:, o:o + n] for bid, o, n in offs})

    def forward(self, *args, **kwargs):
        from ._lib import precision_scope
        with precision_scope(getattr(self, "_cdae_precision", None)):
            return self._forward(*args, **kwargs)

    def _forward(self, x, timesteps, y=None, c=None, x_start=None, z=None, A=None, mask=None):
        """x [N,C,H,W], timesteps [N] -> (eps [N,Cout,H,W] NCHW-contiguous, mu, var, z_post, mask)."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        emb, mu, var, z_post, mask = self.embed(timesteps, y=y, c=c, x_start=x_start, z=z)
        if ops.presplit_ok():
            emb = self._emb_all(emb)
        elif ops.emb_all_train_ok(self):
            emb = EmbAll(emb, ops.emb_all_train(emb, self._emb_flat))      # one GEMM for the 22 emb_layers, fwd and bwd
        hs = []
        h = x.float()
        t16 = ops16.torso16_on() and h.is_cuda     # convert_to_fp16 (reference unet.py:501-507): bf16 activations from the input conv to the head
        for i, module in enumerate(self.input_blocks):
            h = module(h, emb)
            if t16 and i == 0:
                h = ops16.to16(h)
            hs.append(h)
        h = self.middle_block(h, emb)
        for module in self.output_blocks:
            h = module(ops.cat_channels(h, hs.pop()), emb)
        if t16:
            h = ops16.to32(h)                      # the output head normalises and convolves in fp32 (reference unet.py:630 h.type(x.dtype))
        h = self.out[0](h, silu=True, split=True, coef=True)
        return self.out[2](h, out_nchw=True), mu, var, z_post, mask
