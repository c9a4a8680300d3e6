"""Oracle: functional torch-CPU restatement of the CausalDiffAE UNet + causal encoder.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Parameters come in as a plain
dict ``sd`` (state-dict key -> tensor); nothing here is an nn.Module.  Works
with autograd (tests take gradients w.r.t. ``sd`` entries / inputs).

Reference lines restated:
  arch / create_model      script_util.py:119-179, unet.py:388-499
  timestep_embedding       nn.py:551-569
  ResBlock._forward        unet.py:185-198
  AttentionBlock/QKV       unet.py:223-253
  Downsample / Upsample    unet.py:51-105
  UNetModel.forward        unet.py:525-632
  GaussianConvEncoder      nn.py:15-110
  CausalModeling           nn.py:244-312
  reparameterize           nn.py:460-467
"""
import math

import torch
import torch.nn.functional as F

REP_DIM = 512       # script_util.py:13
CONTEXT_DIM = 4     # script_util.py:10
NUM_CLASSES = 10    # script_util.py:9

CHANNEL_MULT = {256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 2, 4, 4), 96: (1, 2, 3, 4),
                64: (1, 2, 3, 4), 32: (1, 2, 2, 2), 28: (1, 2, 2)}      # script_util.py:140-153

ADJ = {  # unet.py:571-578 (+ the commented-out pendulum graph used by the test script)
    "morpho": [[0, 1], [0, 0]],
    "circuit": [[0, 1, 1, 1], [0, 0, 0, 1], [0, 0, 0, 1], [0, 0, 0, 0]],
    "pendulum": [[0, 0, 1, 1], [0, 0, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]],
}


def default_cfg(**over):
    cfg = dict(image_size=64, in_channels=3, num_channels=128, num_res_blocks=2, num_heads=4,
               num_heads_upsample=-1, attention_resolutions="16,8", learn_sigma=False,
               class_cond=False, context_cond=False, rep_cond=False, n_vars=4,
               causal_modeling=False, masking=False, use_scale_shift_norm=True,
               encoder_dims=None, flow_based=False)
    cfg.update(over)
    return cfg


def encoder_dims(image_size, n_vars):
    """Encoder depth so that the last map is 2x2 (fc in-features = dims[-1]*4, nn.py:57).

    Reference quirk Q1: unet.py:377 hard-codes the 6-conv list, which only
    works at 96/128 px; 28/32 px use the reference's 2-var list (nn.py:42-43); 64 px is
    this build's choice (4-var list minus one 64)."""
    if image_size in (28, 32):
        return [16, 32, 64, 128]
    if image_size == 64:
        return [16, 32, 32, 64, 128]
    return [16, 32, 32, 64, 64, 128]


def arch(cfg):
    """Block list of the UNet as (kind, ...) tuples; mirrors unet.py:388-499 walk order."""
    mc = cfg["num_channels"]
    mult = CHANNEL_MULT[cfg["image_size"]]
    att_ds = tuple(cfg["image_size"] // int(r) for r in cfg["attention_resolutions"].split(","))
    heads = cfg["num_heads"]
    heads_up = heads if cfg["num_heads_upsample"] == -1 else cfg["num_heads_upsample"]
    inp = [[("conv", cfg["in_channels"], mc)]]
    chans, ch, ds = [mc], mc, 1
    for lvl, m in enumerate(mult):
        for _ in range(cfg["num_res_blocks"]):
            layers = [("res", ch, m * mc)]
            ch = m * mc
            if ds in att_ds:
                layers.append(("attn", ch, heads))
            inp.append(layers)
            chans.append(ch)
        if lvl != len(mult) - 1:
            inp.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch), ("attn", ch, heads), ("res", ch, ch)]
    out = []
    for lvl, m in list(enumerate(mult))[::-1]:
        for i in range(cfg["num_res_blocks"] + 1):
            layers = [("res", ch + chans.pop(), mc * m)]
            ch = mc * m
            if ds in att_ds:
                layers.append(("attn", ch, heads_up))
            if lvl and i == cfg["num_res_blocks"]:
                layers.append(("up", ch))
                ds //= 2
            out.append(layers)
    c_out = cfg["in_channels"] * (2 if cfg["learn_sigma"] else 1)
    return dict(input=inp, middle=mid, output=out, head_ch=ch, out_channels=c_out, emb=4 * mc)


def _layer_spec(prefix, layer, emb, ssn):
    kind = layer[0]
    if kind == "conv":
        return [(f"{prefix}.weight", (layer[2], layer[1], 3, 3)), (f"{prefix}.bias", (layer[2],))]
    if kind == "res":
        ci, co = layer[1], layer[2]
        s = [(f"{prefix}.in_layers.0.weight", (ci,)), (f"{prefix}.in_layers.0.bias", (ci,)),
             (f"{prefix}.in_layers.2.weight", (co, ci, 3, 3)), (f"{prefix}.in_layers.2.bias", (co,)),
             (f"{prefix}.emb_layers.1.weight", ((2 if ssn else 1) * co, emb)),
             (f"{prefix}.emb_layers.1.bias", ((2 if ssn else 1) * co,)),
             (f"{prefix}.out_layers.0.weight", (co,)), (f"{prefix}.out_layers.0.bias", (co,)),
             (f"{prefix}.out_layers.3.weight", (co, co, 3, 3)), (f"{prefix}.out_layers.3.bias", (co,))]
        if ci != co:
            s += [(f"{prefix}.skip_connection.weight", (co, ci, 1, 1)), (f"{prefix}.skip_connection.bias", (co,))]
        return s
    if kind == "attn":
        c = layer[1]
        return [(f"{prefix}.norm.weight", (c,)), (f"{prefix}.norm.bias", (c,)),
                (f"{prefix}.qkv.weight", (3 * c, c, 1)), (f"{prefix}.qkv.bias", (3 * c,)),
                (f"{prefix}.proj_out.weight", (c, c, 1)), (f"{prefix}.proj_out.bias", (c,))]
    if kind == "down":
        return [(f"{prefix}.op.weight", (layer[1], layer[1], 3, 3)), (f"{prefix}.op.bias", (layer[1],))]
    if kind == "up":
        return [(f"{prefix}.conv.weight", (layer[1], layer[1], 3, 3)), (f"{prefix}.conv.bias", (layer[1],))]
    raise ValueError(kind)


def param_spec(cfg):
    """Ordered (key, shape) list == the reference model's state_dict() layout."""
    a = arch(cfg)
    mc, emb, ssn = cfg["num_channels"], a["emb"], cfg["use_scale_shift_norm"]
    spec = [("time_embed.0.weight", (emb, mc)), ("time_embed.0.bias", (emb,)),
            ("time_embed.2.weight", (emb, emb)), ("time_embed.2.bias", (emb,))]
    if cfg["class_cond"]:
        spec.append(("label_emb.weight", (NUM_CLASSES, emb)))
    if cfg["context_cond"]:
        spec += [("c_emb.0.weight", (256, CONTEXT_DIM)), ("c_emb.0.bias", (256,)),
                 ("c_emb.2.weight", (emb, 256)), ("c_emb.2.bias", (emb,))]
    if cfg["rep_cond"]:
        dims = cfg.get("encoder_dims") or encoder_dims(cfg["image_size"], cfg["n_vars"])
        ci = cfg["in_channels"]
        for i, d in enumerate(dims):
            p = f"rep_emb.encoder.{i}"
            spec += [(f"{p}.0.weight", (d, ci, 3, 3)), (f"{p}.0.bias", (d,)),
                     (f"{p}.1.weight", (d,)), (f"{p}.1.bias", (d,)),
                     (f"{p}.1.running_mean", (d,)), (f"{p}.1.running_var", (d,)),
                     (f"{p}.1.num_batches_tracked", ())]
            ci = d
        spec += [("rep_emb.fc_mu.weight", (REP_DIM, dims[-1] * 4)), ("rep_emb.fc_mu.bias", (REP_DIM,)),
                 ("rep_emb.fc_var.weight", (REP_DIM, dims[-1] * 4)), ("rep_emb.fc_var.bias", (REP_DIM,)),
                 ("up_emb.weight", (emb, REP_DIM)), ("up_emb.bias", (emb,))]
    if cfg["causal_modeling"]:
        nv = cfg["n_vars"]
        d = REP_DIM // nv
        for i in range(nv):
            p = f"causal_mask.nonlinearities.{i}.net"
            spec += [(f"{p}.0.weight", (REP_DIM, d)), (f"{p}.0.bias", (REP_DIM,)),
                     (f"{p}.2.weight", (d, REP_DIM)), (f"{p}.2.bias", (d,))]
    if cfg["flow_based"]:           # MultivariateCausalFlow(dim=2, k=256), nh=100 (unet.py:385-386, nn.py:343-366)
        for net in ("s_cond", "t_cond"):
            for li, (o, i) in zip((0, 2, 4), ((100, 512), (100, 100), (256, 100))):
                spec += [(f"causal_flow.{net}.{li}.weight", (o, i)), (f"causal_flow.{net}.{li}.bias", (o,))]
    for bi, layers in enumerate(a["input"]):
        for li, layer in enumerate(layers):
            spec += _layer_spec(f"input_blocks.{bi}.{li}", layer, emb, ssn)
    for li, layer in enumerate(a["middle"]):
        spec += _layer_spec(f"middle_block.{li}", layer, emb, ssn)
    for bi, layers in enumerate(a["output"]):
        for li, layer in enumerate(layers):
            spec += _layer_spec(f"output_blocks.{bi}.{li}", layer, emb, ssn)
    spec += [("out.0.weight", (a["head_ch"],)), ("out.0.bias", (a["head_ch"],)),
             ("out.2.weight", (a["out_channels"], mc, 3, 3)), ("out.2.bias", (a["out_channels"],))]
    return spec


# ----------------------------------------------------------------------------- layers

def silu(x):
    return x * torch.sigmoid(x)           # nn.py:430-432


def gn32(x, w, b):
    return F.group_norm(x.float(), 32, w, b, 1e-5).type(x.dtype)     # nn.py:435-437,548


def timestep_embedding(t, dim, max_period=10000):
    """nn.py:551-569: [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(P) k / half), fp32."""
    half = dim // 2
    f = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None].float() * f[None]
    e = torch.cat([torch.cos(a), torch.sin(a)], dim=-1)
    if dim % 2:
        e = torch.cat([e, torch.zeros_like(e[:, :1])], dim=-1)
    return e


def resblock(sd, p, x, emb, ssn=True, drop=None):
    """unet.py:185-198.  drop: the training-mode nn.Dropout of out_layers (unet.py:153) as an explicit factor keep_mask / (1 - p)."""
    h = F.conv2d(silu(gn32(x, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"])),
                 sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[:, :, None, None]
    g = gn32  # out_layers.0
    if ssn:
        co = h.shape[1]
        h = g(h, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"]) * (1 + e[:, :co]) + e[:, co:]
    else:
        h = g(h + e, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"])
    h = silu(h)
    if drop is not None:
        h = h * drop
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    k = p + ".skip_connection.weight"
    skip = F.conv2d(x, sd[k], sd[p + ".skip_connection.bias"]) if k in sd else x
    return skip + h


def qkv_attention(qkv, heads):
    """unet.py:239-253 on [B, 3C, T] -> [B, C, T]; per head the channel order is q|k|v."""
    b, c3, T = qkv.shape
    q3 = qkv.reshape(b * heads, c3 // heads, T)
    ch = q3.shape[1] // 3
    q, k, v = q3[:, :ch], q3[:, ch:2 * ch], q3[:, 2 * ch:]
    s = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * s, k * s)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    return torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, T)


def attnblock(sd, p, x, heads):
    """unet.py:223-231."""
    b, c = x.shape[:2]
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(gn32(xf, sd[p + ".norm.weight"], sd[p + ".norm.bias"]), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    h = qkv_attention(qkv, heads)
    h = F.conv1d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + h).reshape(x.shape)


def run_layer(sd, p, layer, h, emb, ssn=True):
    kind = layer[0]
    if kind == "conv":
        return F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
    if kind == "res":
        return resblock(sd, p, h, emb, ssn)
    if kind == "attn":
        return attnblock(sd, p, h, layer[2])
    if kind == "down":
        return F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)       # unet.py:99
    if kind == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")                                   # unet.py:76
        return F.conv2d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
    raise ValueError(kind)


# ----------------------------------------------------------------------------- causal encoder

def encode(sd, x, n_layers, training=False, new_stats=None):
    """nn.py:93-110: [conv3x3 s2 -> BatchNorm2d -> LeakyReLU(.01)] x L, flatten, fc_mu, softplus(fc_var)+1e-8.

    training=True uses batch statistics (biased var) like nn.BatchNorm2d.train(); if `new_stats`
    is a dict it receives the updated running stats (momentum 0.1, unbiased var)."""
    h = x
    for i in range(n_layers):
        p = f"rep_emb.encoder.{i}"
        h = F.conv2d(h, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=2, padding=1)
        if training:
            mean = h.mean(dim=(0, 2, 3))
            var = h.var(dim=(0, 2, 3), unbiased=False)
            if new_stats is not None:
                n = h.numel() / h.shape[1]
                new_stats[p + ".1.running_mean"] = 0.9 * sd[p + ".1.running_mean"] + 0.1 * mean.detach()
                new_stats[p + ".1.running_var"] = 0.9 * sd[p + ".1.running_var"] + 0.1 * var.detach() * n / (n - 1)
        else:
            mean, var = sd[p + ".1.running_mean"], sd[p + ".1.running_var"]
        h = (h - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5)
        h = h * sd[p + ".1.weight"][None, :, None, None] + sd[p + ".1.bias"][None, :, None, None]
        h = F.leaky_relu(h, 0.01)
    h = h.flatten(1)
    mu = F.linear(h, sd["rep_emb.fc_mu.weight"], sd["rep_emb.fc_mu.bias"])
    var = F.softplus(F.linear(h, sd["rep_emb.fc_var.weight"], sd["rep_emb.fc_var.bias"])) + 1e-8
    return mu, var


def n_encoder_layers(sd):
    return len({k.split(".")[2] for k in sd if k.startswith("rep_emb.encoder.")})


def causal_masking(u, A, n_vars):
    """nn.py:290-295: z_pre_i = sum_{j in pa(i)} u_j  (A^T @ u on [N, nv, d])."""
    return torch.matmul(A.t(), u.reshape(u.shape[0], n_vars, -1))


def nonlinearity_add_back_noise(sd, u, z_pre, n_vars):
    """nn.py:297-312: z_post_i = MLP_i(z_pre_i) + u_i, MLP_i = Linear(d,512) LeakyReLU Linear(512,d)."""
    u3 = u.reshape(u.shape[0], n_vars, -1)
    outs = []
    for i in range(n_vars):
        p = f"causal_mask.nonlinearities.{i}.net"
        hmid = F.leaky_relu(F.linear(z_pre[:, i], sd[p + ".0.weight"], sd[p + ".0.bias"]), 0.01)
        outs.append(F.linear(hmid, sd[p + ".2.weight"], sd[p + ".2.bias"]) + u3[:, i])
    return torch.stack(outs, dim=1).reshape(u.shape[0], -1)


def _flow_cond(sd, net, x):
    p = f"causal_flow.{net}"
    h = F.relu(F.linear(x, sd[p + ".0.weight"], sd[p + ".0.bias"]))
    h = F.relu(F.linear(h, sd[p + ".2.weight"], sd[p + ".2.bias"]))
    return torch.sigmoid(F.linear(h, sd[p + ".4.weight"], sd[p + ".4.bias"]))


def _flow_mask(C, i, k):
    col = C[:, i]
    return col.repeat(k, 1).T.reshape(-1) if bool((col == 1).any()) else torch.zeros(C.shape[0] * k)


def causal_flow(sd, e, C, dim=2, k=256):
    """nn.py:368-393: z_i = exp(s_i) e_i + t_i, conditioners see the blocks filled so far times column i of C."""
    N = e.shape[0]
    e3 = e.reshape(N, dim, k)
    z = torch.zeros(N, dim, k)
    log_det = torch.zeros(N)
    for i in range(dim):
        inp = z.reshape(N, dim * k) * _flow_mask(C, i, k)
        s, t = _flow_cond(sd, "s_cond", inp), _flow_cond(sd, "t_cond", inp)
        z = torch.cat([z[:, :i], (torch.exp(s) * e3[:, i] + t)[:, None], z[:, i + 1:]], dim=1)
        log_det = log_det + s.sum(dim=1)
    return z.reshape(N, dim * k), log_det


def causal_flow_reverse(sd, z, C, dim=2, k=256):
    """nn.py:395-426: conditioners see the FULL z (so this is not the inverse of `causal_flow` when diag(C) != 0)."""
    N = z.shape[0]
    z3 = z.reshape(N, dim, k)
    log_det = torch.zeros(N)
    es = []
    for i in range(dim):
        inp = z.reshape(N, dim * k) * _flow_mask(C, i, k)
        s, t = _flow_cond(sd, "s_cond", inp), _flow_cond(sd, "t_cond", inp)
        es.append(torch.exp(-s) * (z3[:, i] - t))
        log_det = log_det - s.sum(dim=1)
    e = torch.cat(es, dim=1)
    log_prob = -0.5 * ((e - 1.0) ** 2).sum(dim=1) - 0.5 * dim * k * math.log(2 * math.pi)
    return log_det, log_prob


def reparameterize(m, v, eps):
    """nn.py:460-467 with the normal draw `eps` injected."""
    return m + (v ** 0.5) * eps


# ----------------------------------------------------------------------------- full forward

def unet_forward(sd, cfg, x, t, y=None, c=None, x_start=None, z=None, A=None,
                 eps_z=None, cfg_mask=None, training=False, new_stats=None):
    """unet.py:525-632.  `t` is the (already rescaled) float timestep the model sees.

    Random draws are injected: eps_z ~ N(0,I) [N,512] (reparameterize), cfg_mask in {0,1}[N]
    (classifier-free masking, unet.py:600-613).  A defaults to the graph hard-coded in the
    reference forward (n_vars==2 -> morpho, else circuit); returns (eps, mu, var, z_post, mask)."""
    a = arch(cfg)
    ssn = cfg["use_scale_shift_norm"]
    emb = timestep_embedding(t, cfg["num_channels"])
    emb = F.linear(silu(F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])),
                   sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    assert (y is not None) == cfg["class_cond"]
    if cfg["class_cond"]:
        emb = emb + sd["label_emb.weight"][y]
    if cfg["context_cond"]:
        emb = emb + F.linear(silu(F.linear(c, sd["c_emb.0.weight"], sd["c_emb.0.bias"])),
                             sd["c_emb.2.weight"], sd["c_emb.2.bias"])
    mu = var = z_post = mask = None
    if cfg["rep_cond"]:
        if z is None:
            mu, var = encode(sd, x_start, n_encoder_layers(sd), training, new_stats)
            if cfg["causal_modeling"]:
                nv = cfg["n_vars"]
                if A is None:
                    A = torch.tensor(ADJ["morpho"] if nv == 2 else ADJ["circuit"], dtype=torch.float32)
                if cfg["flow_based"]:                                   # unet.py:580-587
                    Cm = torch.eye(A.shape[0]) - A
                    z_post, _ = causal_flow(sd, mu, Cm)
                    mask = -torch.mean(causal_flow_reverse(sd, z_post, Cm)[0])
                else:
                    z_pre = causal_masking(mu, A, nv)
                    z_post = nonlinearity_add_back_noise(sd, mu, z_pre, nv)
                z = reparameterize(z_post, var * 0.001, eps_z)         # unet.py:592
            else:
                z = reparameterize(mu, var * 0.001, eps_z)
            if cfg["masking"]:
                mask = cfg_mask
                z = z * mask[:, None]
                z_post = z_post * mask[:, None]
        emb = emb + F.linear(z, sd["up_emb.weight"], sd["up_emb.bias"])
    hs, h = [], x
    for bi, layers in enumerate(a["input"]):
        for li, layer in enumerate(layers):
            h = run_layer(sd, f"input_blocks.{bi}.{li}", layer, h, emb, ssn)
        hs.append(h)
    for li, layer in enumerate(a["middle"]):
        h = run_layer(sd, f"middle_block.{li}", layer, h, emb, ssn)
    for bi, layers in enumerate(a["output"]):
        h = torch.cat([h, hs.pop()], dim=1)
        for li, layer in enumerate(layers):
            h = run_layer(sd, f"output_blocks.{bi}.{li}", layer, h, emb, ssn)
    h = F.conv2d(silu(gn32(h, sd["out.0.weight"], sd["out.0.bias"])), sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return h, mu, var, z_post, mask
