#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

This is the only file in the repo that imports /root/reference.  It never
travels as a dependency: tests read the committed .npz files.  Weights and
inputs are closed-form (oracle/closed_form.py: hash of key name + flat index),
so fixtures hold expected OUTPUTS plus the small amount of metadata needed to
regenerate the inputs.

    python tools/gen_golden.py [--only G1,G6] [--out tests/golden]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch as th

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from improved_diffusion import gaussian_diffusion as gd            # noqa: E402  (reference)
from improved_diffusion import nn as rnn                           # noqa: E402
from improved_diffusion import respace as rrespace                 # noqa: E402
from improved_diffusion import script_util as rsu                  # noqa: E402
from improved_diffusion import unet as runet                       # noqa: E402

from oracle.closed_form import fill_value, fill_value_trained, synth, synth_noise   # noqa: E402
from oracle.unet_ref import ADJ, encoder_dims                      # noqa: E402

th.set_grad_enabled(True)


def probe(t):
    """Compact fingerprint of a big tensor: head, strided sample, sum, sum of squares."""
    f = t.detach().double().flatten()
    return dict(head=f[:64].float().numpy(), strided=f[::997].float().numpy(),
                sum=np.float64(f.sum().item()), sumsq=np.float64((f * f).sum().item()))


def flat_probe(prefix, t, out):
    for k, v in probe(t).items():
        out[f"{prefix}/{k}"] = v


def load_closed_form(module, prefix=""):
    sd = module.state_dict()
    for k in sd:
        sd[k] = fill_value(prefix + k, sd[k].shape)
    module.load_state_dict(sd)
    return module


def build_model(name):
    """Reference model + diffusion for the BASELINE shapes (M32 / P64 / C64 / tiny)."""
    base = rsu.model_and_diffusion_defaults()
    cfgs = {
        "M32": dict(image_size=32, in_channels=1, n_vars=2, class_cond=True),
        "P64": dict(image_size=64, in_channels=4, n_vars=4),
        "C64": dict(image_size=64, in_channels=3, n_vars=4),
        "T28": dict(image_size=28, in_channels=1, n_vars=2, class_cond=True, num_channels=32, num_res_blocks=1),
    }
    over = dict(rep_cond=True, causal_modeling=True)
    over.update(cfgs[name])
    return over, base


def make(name, respacing="", masking=False, **extra):
    over, base = build_model(name)
    base.update(over)
    base["timestep_respacing"] = respacing
    base["masking"] = masking
    base.update(extra)
    model, diff = rsu.create_model_and_diffusion(**base)
    # Q1: the committed encoder depth only works at 96/128 px; use the class's own hidden_dims kwarg.
    if base["rep_cond"]:
        dims = encoder_dims(base["image_size"], base["n_vars"])
        model.rep_emb = rnn.GaussianConvEncoder(base["in_channels"], 512, hidden_dims=dims, num_vars=base["n_vars"])
    load_closed_form(model)
    return model, diff, base


# --------------------------------------------------------------------------- G1
def g1_schedules(out_dir):
    out = {}
    for rs in ["", "ddim100", "ddim250", "250", "100", "ddim50", "10,10,10"]:
        d = rsu.create_gaussian_diffusion(steps=1000, timestep_respacing=rs, rescale_timesteps=True)
        tag = rs or "full"
        for n in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                  "posterior_mean_coef1", "posterior_mean_coef2"]:
            out[f"{tag}/{n}"] = getattr(d, n)
        out[f"{tag}/timestep_map"] = np.array(d.timestep_map, dtype=np.int64)
        out[f"{tag}/fixed_large_variance"] = np.append(d.posterior_variance[1], d.betas[1:])
    d = rsu.create_gaussian_diffusion(steps=500, noise_schedule="cosine", timestep_respacing="")
    out["cosine500/betas"] = d.betas
    cases = [(300, [10, 15, 20]), (1000, "ddim25"), (1000, "1000"), (1000, "7"), (1000, "333,1"), (50, "ddim10"),
             (1000, "1,1")]
    for T, spec in cases:
        key = f"space/{T}/{spec if isinstance(spec, str) else ','.join(map(str, spec))}"
        out[key] = np.array(sorted(rrespace.space_timesteps(T, spec)), dtype=np.int64)
    errs = []
    for T, spec in [(1000, "ddim7"), (10, "20"), (1000, "ddim999")]:
        try:
            rrespace.space_timesteps(T, spec)
            errs.append(0)
        except ValueError:
            errs.append(1)
    out["space/errors"] = np.array(errs)
    np.savez_compressed(os.path.join(out_dir, "g1_schedules.npz"), **out)


# --------------------------------------------------------------------------- G2
def g2_temb(out_dir):
    out = {}
    t = th.tensor([0.0, 1.0, 10.0, 249.0, 990.0, 999.0, 123.5])
    out["temb128"] = rnn.timestep_embedding(t, 128).numpy()
    out["temb33"] = rnn.timestep_embedding(t, 33).numpy()
    out["t"] = t.numpy()
    d = rsu.create_gaussian_diffusion(steps=1000, timestep_respacing="ddim100", rescale_timesteps=True)
    seen = {}
    wm = d._wrap_model(lambda x, ts, **kw: seen.setdefault("ts", ts))
    ts = th.tensor([0, 1, 50, 99], dtype=th.int64)
    wm(None, ts)
    out["wrapped_in"] = ts.numpy()
    out["wrapped_out"] = seen["ts"].numpy()
    d2 = rsu.create_gaussian_diffusion(steps=1000, timestep_respacing="250", rescale_timesteps=False)
    seen.clear()
    d2._wrap_model(lambda x, ts, **kw: seen.setdefault("ts", ts))(None, th.tensor([0, 248, 249]))
    out["wrapped_norescale_out"] = seen["ts"].numpy()
    np.savez_compressed(os.path.join(out_dir, "g2_temb.npz"), **out)


# --------------------------------------------------------------------------- G3
def g3_blocks(out_dir):
    out, meta = {}, {}

    def run_block(tag, mod, x_shape, emb=True, extra=None):
        load_closed_form(mod, tag + ".")
        x = synth(tag + ".x", x_shape).requires_grad_(True)
        args = [x]
        if emb:
            e = synth(tag + ".emb", (x_shape[0], 512)).requires_grad_(True)
            args.append(e)
        y = mod(*args)
        gy = synth(tag + ".gy", tuple(y.shape))
        (y * gy).sum().backward()
        out[f"{tag}/y"] = y.detach().numpy()
        out[f"{tag}/gx"] = x.grad.numpy()
        if emb:
            out[f"{tag}/gemb"] = e.grad.numpy()
        for k, p in mod.named_parameters():
            flat_probe(f"{tag}/g.{k}", p.grad, out)
        meta[tag] = dict(x_shape=list(x_shape), y_shape=list(y.shape), emb=emb, **(extra or {}))

    run_block("res_same", runet.ResBlock(128, 512, 0.0, use_scale_shift_norm=True), (2, 128, 4, 4))
    run_block("res_skip", runet.ResBlock(128, 512, 0.0, out_channels=256, use_scale_shift_norm=True), (2, 128, 4, 4))
    run_block("res_cat", runet.ResBlock(384, 512, 0.0, out_channels=128, use_scale_shift_norm=True), (2, 384, 4, 4))
    run_block("res_nossn", runet.ResBlock(64, 512, 0.0, out_channels=96, use_scale_shift_norm=False), (2, 64, 6, 6))
    for ch, T, heads in [(96, 256, 4), (128, 64, 4), (64, 256, 4), (64, 16, 4)]:
        s = int(T ** 0.5)
        run_block(f"attn_{ch}_{T}", runet.AttentionBlock(ch * heads, num_heads=heads), (1, ch * heads, s, s), emb=False,
                  extra=dict(heads=heads))
    run_block("down", runet.Downsample(128, True), (2, 128, 8, 8), emb=False)
    run_block("up", runet.Upsample(128, True), (2, 128, 4, 4), emb=False)
    # bare QKVAttention (unet.py:239-253) on a [B*H, 3*ch, T] tensor
    qkv = synth("qkv.x", (8, 3 * 32, 64), -2, 2).requires_grad_(True)
    y = runet.QKVAttention()(qkv)
    gy = synth("qkv.gy", tuple(y.shape))
    (y * gy).sum().backward()
    out["qkv/y"], out["qkv/gx"] = y.detach().numpy(), qkv.grad.numpy()
    # output head = GN32 -> SiLU -> conv3x3 (unet.py:495-499)
    head = th.nn.Sequential(rnn.normalization(128), rnn.SiLU(), rnn.conv_nd(2, 128, 4, 3, padding=1))
    run_block("head", head, (2, 128, 8, 8), emb=False)
    np.savez_compressed(os.path.join(out_dir, "g3_blocks.npz"), **out)
    with open(os.path.join(out_dir, "g3_blocks.json"), "w") as f:
        json.dump(meta, f, indent=1)


# --------------------------------------------------------------------------- G4
def g4_encoder(out_dir):
    out = {}
    for tag, C, S, nv in [("enc32", 1, 32, 2), ("enc64", 4, 64, 4), ("enc96", 4, 96, 4)]:
        dims = encoder_dims(S, nv)
        enc = rnn.GaussianConvEncoder(C, 512, hidden_dims=dims, num_vars=nv)
        load_closed_form(enc, "rep_emb.")
        # pick an input on which no LeakyReLU pre-activation (train-mode BatchNorm output) sits near the kink: the branch taken at
        # |pre| ~ 1e-7 is decided by fp32 rounding and would need an exemption in the parity test
        saved = {k: v.clone() for k, v in enc.state_dict().items()}
        for salt in ["", "b", "c", "d", "e", "f"]:
            name = tag + salt + ".x"
            x = synth(name, (4, C, S, S), 0.0, 1.0)
            enc.train()
            h, closest = x, 1e9
            with th.no_grad():
                for layer in enc.encoder:
                    pre = layer[1](layer[0](h))
                    closest = min(closest, float(pre.abs().min()))
                    h = layer[2](pre)
            enc.load_state_dict(saved)                                # the probe pass must not leave running statistics behind
            if closest > 2e-6:
                break
        else:
            raise RuntimeError("no kink-free encoder input found for " + tag)
        out[f"{tag}/input_name"] = np.array(name)
        out[f"{tag}/closest_preactivation"] = np.float64(closest)
        enc.eval()
        mu, var = enc.encode(x)
        out[f"{tag}/eval_mu"], out[f"{tag}/eval_var"] = mu.detach().numpy(), var.detach().numpy()
        enc.train()
        xg = x.clone().requires_grad_(True)
        mu, var = enc.encode(xg)
        out[f"{tag}/train_mu"], out[f"{tag}/train_var"] = mu.detach().numpy(), var.detach().numpy()
        gmu, gvar = synth(tag + ".gmu", (4, 512)), synth(tag + ".gvar", (4, 512))
        ((mu * gmu).sum() + (var * gvar).sum()).backward()
        out[f"{tag}/train_gx"] = xg.grad.numpy()
        for k, p in enc.named_parameters():
            if p.grad is not None:
                flat_probe(f"{tag}/g.{k}", p.grad, out)
        for k, v in enc.state_dict().items():
            if "running" in k:
                out[f"{tag}/after.{k}"] = v.numpy()
    for nv, graphs in [(2, ["morpho"]), (4, ["circuit", "pendulum"])]:
        cm = rnn.CausalModeling(latent_dim=512, num_var=nv, learn=False)
        load_closed_form(cm, "causal_mask.")
        u = synth(f"causal{nv}.u", (3, 512)).requires_grad_(True)
        for g in graphs:
            A = th.tensor(ADJ[g], dtype=th.float32)
            z_pre = cm.causal_masking(u, A)
            z_post = cm.nonlinearity_add_back_noise(u, z_pre)
            out[f"causal/{g}/z_pre"], out[f"causal/{g}/z_post"] = z_pre.detach().numpy(), z_post.detach().numpy()
        gz = synth(f"causal{nv}.gz", (3, 512))
        (z_post * gz).sum().backward()
        out[f"causal/{graphs[-1]}/gu"] = u.grad.numpy()
    th.manual_seed(7)
    m, v = synth("rep.m", (3, 512)), synth("rep.v", (3, 512), 0.01, 1.0)
    z = rnn.reparameterize(m, v)
    th.manual_seed(7)
    out["reparam/eps"] = th.randn(m.size()).numpy()
    out["reparam/z"] = z.numpy()
    np.savez_compressed(os.path.join(out_dir, "g4_encoder.npz"), **out)


# --------------------------------------------------------------------------- G5
def g5_rep_loss(out_dir):
    out = {}
    d = rsu.create_gaussian_diffusion(steps=1000)
    for nv in (2, 4):
        N = 5
        mu, var = synth(f"rl{nv}.mu", (N, 512)), synth(f"rl{nv}.var", (N, 512), 0.05, 2.0)
        zp, c = synth(f"rl{nv}.zp", (N, 512)), synth(f"rl{nv}.c", (N, nv), 0.0, 1.0)
        mask = th.tensor([1.0, 0.0, 1.0, 1.0, 0.0])
        out[f"nv{nv}/causal"] = d.representation_loss(mu, var, zp, True, None, c).numpy()
        out[f"nv{nv}/plain"] = d.representation_loss(mu, var, zp, False, None, c).numpy()
        out[f"nv{nv}/causal_masked"] = d.representation_loss(mu, var, zp, True, mask, c).numpy()
    np.savez_compressed(os.path.join(out_dir, "g5_rep_loss.npz"), **out)


# --------------------------------------------------------------------------- G6
def model_inputs(tag, base, N):
    C, S, nv = base["in_channels"], base["image_size"], base["n_vars"]
    x = synth(tag + ".x", (N, C, S, S))
    x0 = synth(tag + ".x0", (N, C, S, S), 0.0, 1.0)
    c = synth(tag + ".c", (N, nv), 0.0, 1.0)
    z = synth(tag + ".z", (N, 512))
    y = th.tensor([(3 * i + 1) % 10 for i in range(N)], dtype=th.int64) if base["class_cond"] else None
    return x, x0, c, z, y


def g6_unet(out_dir):
    out = {}
    N = 2
    for tag in ["M32", "P64", "C64"]:
        model, diff, base = make(tag)
        model.eval()
        x, x0, c, z, y = model_inputs(tag, base, N)
        t = th.tensor([37.0, 990.0])
        kw = dict(y=y) if y is not None else {}
        with th.no_grad():
            e, *_ = model(x, t, z=z, **kw)                     # sampling path (z given)
            out[f"{tag}/eps_z"] = e.numpy()
            th.manual_seed(11)
            e2, mu, var, zp, mask = model(x, t, x_start=x0, **kw)      # encoder path, eval-mode BN
            th.manual_seed(11)
            out[f"{tag}/eps_draw"] = th.randn(N, 512).numpy()
            out[f"{tag}/eps_enc"], out[f"{tag}/mu"], out[f"{tag}/var"], out[f"{tag}/z_post"] = (
                e2.numpy(), mu.numpy(), var.numpy(), zp.numpy())
        out[f"{tag}/keys"] = np.array(list(model.state_dict().keys()))
        out[f"{tag}/shapes"] = np.array([str(tuple(v.shape)) for v in model.state_dict().values()])
        out[f"{tag}/n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
    np.savez_compressed(os.path.join(out_dir, "g6_unet.npz"), **out)


# --------------------------------------------------------------------------- G7
def kl_weight_at(step, total=50000):
    # restated from reference train_util.py:176-187 (train_util itself cannot be imported: blobfile/mpi4py absent)
    if step >= total:
        return 1.0
    if step <= 0:
        return 0.0
    return step / (total - 1)


def g7_train(out_dir):
    from torch.optim import AdamW
    import copy
    out = {}
    for variant, masking in [("plain", False), ("masked", True)]:
        model, diff, base = make("T28", masking=masking)
        model.train()
        N = 4
        params = list(model.parameters())
        ema = copy.deepcopy(params)
        opt = AdamW(params, lr=1e-4, weight_decay=0.0)
        names = [k for k, _ in model.named_parameters()]
        sel = [0, 3, len(names) // 2, len(names) - 1, names.index("rep_emb.encoder.0.0.weight"),
               names.index("causal_mask.nonlinearities.1.net.2.weight"), names.index("input_blocks.1.0.in_layers.2.weight"),
               names.index("out.2.weight")]
        out[f"{variant}/sel_names"] = np.array([names[i] for i in sel])
        diff.kl_weight = 0.0
        for step in range(3):
            # NB kl_weight would be ~2e-5*step at these steps; use a visible weight so the KL path is exercised.
            diff.kl_weight = [0.0, 0.25, 0.5][step]
            x0 = synth(f"T28.{step}.x0", (N, 1, 28, 28), 0.0, 1.0)
            c = synth(f"T28.{step}.c", (N, 2), 0.0, 1.0)
            y = th.tensor([(step + 2 * i) % 10 for i in range(N)], dtype=th.int64)
            t = th.tensor([(137 * (step + 1) + 251 * i) % 1000 for i in range(N)], dtype=th.int64)
            noise = synth(f"T28.{step}.noise", (N, 1, 28, 28), -1.7, 1.7)
            for p in params:
                p.grad = None
            th.manual_seed(100 + step)
            terms = diff.training_losses(model, x0, t, model_kwargs=dict(c=c, y=y), noise=noise, rep_cond=True,
                                         causal_modeling=True)
            th.manual_seed(100 + step)
            out[f"{variant}/step{step}/eps_draw"] = th.randn(N, 512).numpy()
            if masking:
                out[f"{variant}/step{step}/cfg_mask"] = th.bernoulli(th.zeros(N) + 0.5).numpy()
            weights = th.ones(N)
            (terms["loss"] * weights).mean().backward()
            for k in ("loss", "mse", "kld_rep"):
                out[f"{variant}/step{step}/{k}"] = terms[k].detach().numpy()
            out[f"{variant}/step{step}/t"] = t.numpy()
            out[f"{variant}/step{step}/grad_sqsum"] = np.float64(sum((p.grad.double() ** 2).sum().item() for p in params))
            if step == 0:
                for i in sel:
                    flat_probe(f"{variant}/grad0/{names[i]}", params[i].grad, out)
            opt.step()
            rnn.update_ema(ema, params, rate=0.9999)
            if step in (0, 2):
                for i in sel:
                    flat_probe(f"{variant}/after{step + 1}/{names[i]}", params[i], out)
                    flat_probe(f"{variant}/ema{step + 1}/{names[i]}", ema[i], out)
                out[f"{variant}/after{step + 1}/param_sum"] = np.float64(sum(p.double().sum().item() for p in params))
                bn = model.state_dict()["rep_emb.encoder.1.1.running_var"]
                out[f"{variant}/after{step + 1}/bn_running_var"] = bn.numpy().copy()
    out["kl_weight_sched"] = np.array([kl_weight_at(s) for s in (0, 1, 2, 25000, 49999, 50000, 60000)])
    np.savez_compressed(os.path.join(out_dir, "g7_train.npz"), **out)


# --------------------------------------------------------------------------- G8
def g8_ddim(out_dir):
    out = {}
    N = 2
    model, diff, base = make("P64", respacing="ddim100")
    model.eval()
    x, x0, c, z, _ = model_inputs("P64", base, N)
    A = th.tensor(ADJ["pendulum"], dtype=th.float32)
    with th.no_grad():
        # counterfactual pattern of image_causaldae_test.py:535-594 (pendulum branch), T'=100
        mu, var = model.rep_emb.encode(x0)
        var = th.ones_like(mu) * 0.001
        z_pre = model.causal_mask.causal_masking(mu, A)
        z_post = model.causal_mask.nonlinearity_add_back_noise(mu, z_pre)
        z_post[:, :128] = 0.2
        th.manual_seed(5)
        zz = rnn.reparameterize(z_post, var)
        th.manual_seed(5)
        out["cf/eps_draw"] = th.randn(N, 512).numpy()
        out["cf/z"] = zz.numpy()
        noise = synth("P64.qnoise", (N, 4, 64, 64), -1.7, 1.7)
        t = th.full((N,), 99, dtype=th.int64)
        x_t = diff.q_sample(x0, t, noise=noise)
        out["cf/x_t"] = x_t.numpy()
        # single DDIM step + single ancestral step at t = 99 and t = 0
        for tv in (99, 0):
            tt = th.full((N,), tv, dtype=th.int64)
            o = diff.ddim_sample(model, x_t, tt, model_kwargs=dict(z=zz))
            out[f"ddim_step{tv}/sample"], out[f"ddim_step{tv}/pred_xstart"] = o["sample"].numpy(), o["pred_xstart"].numpy()
            th.manual_seed(21)
            o = diff.ddim_sample(model, x_t, tt, model_kwargs=dict(z=zz), eta=0.7)
            th.manual_seed(21)
            out[f"ddim_eta_step{tv}/noise"] = th.randn(N, 4, 64, 64).numpy()
            out[f"ddim_eta_step{tv}/sample"] = o["sample"].numpy()
            th.manual_seed(22)
            o = diff.p_sample(model, x_t, tt, model_kwargs=dict(z=zz))
            th.manual_seed(22)
            out[f"p_step{tv}/noise"] = th.randn(N, 4, 64, 64).numpy()
            out[f"p_step{tv}/sample"], out[f"p_step{tv}/pred_xstart"] = o["sample"].numpy(), o["pred_xstart"].numpy()
        # full DDIM-100 loop, eta = 0 (the drawn noise is multiplied by sigma = 0)
        k = 0
        for o in diff.ddim_sample_loop_progressive(model, (N, 4, 64, 64), noise=x_t, model_kwargs=dict(z=zz)):
            k += 1
            if k in (1, 2, 10, 50, 100):
                out[f"loop/sample_after{k}"] = o["sample"].numpy()
    np.savez_compressed(os.path.join(out_dir, "g8_ddim.npz"), **out)


# --------------------------------------------------------------------------- G9
def g9_vlb(out_dir):
    """Learned-sigma / variational-bound branches (gaussian_diffusion.py:289-303,682-715,792-837,862-931; losses.py)."""
    from improved_diffusion import losses as rloss
    out = {}
    # (a) losses.py on closed-form inputs
    sh = (3, 2, 8, 8)
    m1, lv1 = synth("G9.m1", sh, -1.0, 1.0), synth("G9.lv1", sh, -6.0, 0.5)
    m2, lv2 = synth("G9.m2", sh, -1.0, 1.0), synth("G9.lv2", sh, -6.0, 0.5)
    out["losses/normal_kl"] = rloss.normal_kl(m1, lv1, m2, lv2).numpy()
    out["losses/normal_kl_scalar"] = rloss.normal_kl(m1, lv1, 0.0, 0.0).numpy()
    xq = (th.round(synth("G9.xq", sh, 0.0, 255.0)) / 127.5 - 1.0)
    xq.view(-1)[:7] = th.tensor([-1.0, 1.0, -1.0, 1.0, 0.9992, -0.9992, 0.0])
    ls = synth("G9.ls", sh, -4.0, 0.0)
    out["losses/xq"] = xq.numpy()
    out["losses/dgll"] = rloss.discretized_gaussian_log_likelihood(xq, means=m2, log_scales=ls).numpy()
    out["losses/cdf"] = rloss.approx_standard_normal_cdf(synth("G9.cdf", (64,), -5.0, 5.0)).numpy()

    N = 4
    t = th.tensor([0, 5, 500, 999], dtype=th.int64)
    x0 = th.round(synth("G9.x0", (N, 1, 28, 28), 0.0, 255.0)) / 127.5 - 1.0
    noise = synth("G9.noise", (N, 1, 28, 28), -1.7, 1.7)
    c = synth("G9.c", (N, 2), 0.0, 1.0)
    y = th.tensor([1, 3, 5, 7], dtype=th.int64)
    z = synth("G9.z", (N, 512), -1.0, 1.0)
    out["t"] = t.numpy()
    for tag, extra in [("range", dict(learn_sigma=True)), ("fixed", dict()), ("xstart", dict(learn_sigma=True, predict_xstart=True)),
                       ("small", dict(sigma_small=True))]:
        model, diff, base = make("T28", **extra)
        model.eval()
        kw = dict(c=c, y=y, z=z)
        x_t = diff.q_sample(x0, t, noise=noise)
        with th.no_grad():
            for clip in (True, False):
                pm = diff.p_mean_variance(model, x_t, t, clip_denoised=clip, model_kwargs=kw)
                for k, v in pm.items():
                    out[f"{tag}/pmv_clip{int(clip)}/{k}"] = v.numpy()
                vb = diff._vb_terms_bpd(model, x0, x_t, t, clip_denoised=clip, model_kwargs=kw)
                out[f"{tag}/vb_clip{int(clip)}/output"] = vb["output"].numpy()
                out[f"{tag}/vb_clip{int(clip)}/pred_xstart"] = vb["pred_xstart"].numpy()
            th.manual_seed(77)
            ps = diff.p_sample(model, x_t, t, model_kwargs=kw)
            th.manual_seed(77)
            out[f"{tag}/p_sample/noise"] = th.randn_like(x_t).numpy()
            out[f"{tag}/p_sample/sample"] = ps["sample"].numpy()
            dd = diff.ddim_sample(model, x_t, t, model_kwargs=kw, eta=0.0)
            out[f"{tag}/ddim/sample"] = dd["sample"].numpy()
            out[f"{tag}/ddim/pred_xstart"] = dd["pred_xstart"].numpy()
            out[f"{tag}/prior_bpd"] = diff._prior_bpd(x0).numpy()
        # vb gradient with respect to the raw model output (both halves), per-sample weights 1..N
        with th.no_grad():
            raw = model(x_t, diff._scale_timesteps(t), **kw)[0]
        raw = raw.clone().requires_grad_(True)
        vb = diff._vb_terms_bpd(lambda *a, r=raw, **k: (r, None, None, None, None), x0, x_t, t, clip_denoised=False)["output"]
        (vb * th.arange(1, N + 1, dtype=th.float32)).sum().backward()
        out[f"{tag}/vb_grad/raw"] = raw.detach().numpy()
        out[f"{tag}/vb_grad/draw"] = raw.grad.numpy()

    # (c) hybrid loss as upstream improved-diffusion intends it.  The committed training_losses crashes for learn_sigma=True: its
    # frozen-output lambda returns one tensor where p_mean_variance unpacks five (gaussian_diffusion.py:286,826) - so the terms are
    # composed here from the reference's own pieces with a five-tuple lambda.
    model, diff, base = make("T28", learn_sigma=True)
    model.train()
    params = list(model.parameters())
    names = [k for k, _ in model.named_parameters()]
    x_t = diff.q_sample(x0, t, noise=noise)
    kw = dict(c=c, y=y, x_start=x0)
    th.manual_seed(5)
    model_output, mu, var, z_post, mask = model(x_t, diff._scale_timesteps(t), **kw)
    th.manual_seed(5)
    out["hybrid/eps_draw"] = th.randn(N, 512).numpy()
    kld = diff.representation_loss(mu, var, z_post, True, mask, c)
    eps_out, var_out = th.split(model_output, 1, dim=1)
    frozen = th.cat([eps_out.detach(), var_out], dim=1)
    vb = diff._vb_terms_bpd(model=lambda *a, r=frozen, **k: (r, None, None, None, None), x_start=x0, x_t=x_t, t=t, clip_denoised=False)["output"]
    vb = vb * (diff.num_timesteps / 1000.0)
    mse = rnn.mean_flat((noise - eps_out) ** 2)
    loss = mse + vb                      # reference :849-850: with "vb" present the representation KL is NOT added
    loss.mean().backward()
    for k, v in (("mse", mse), ("vb", vb), ("loss", loss), ("kld_rep", kld)):
        out[f"hybrid/{k}"] = v.detach().numpy()
    out["hybrid/grad_sqsum"] = np.float64(sum((p.grad.double() ** 2).sum().item() for p in params if p.grad is not None))
    for nme in ("out.2.weight", "out.2.bias", "input_blocks.1.0.in_layers.2.weight", "time_embed.0.weight"):
        flat_probe(f"hybrid/grad/{nme}", params[names.index(nme)].grad, out)

    # (d) pure-bound training loss (use_kl=True -> RESCALED_KL, fixed variance): the reference's own training_losses runs
    model, diff, base = make("T28", use_kl=True)
    model.train()
    params = list(model.parameters())
    terms = diff.training_losses(model, x0, t, model_kwargs=dict(c=c, y=y, z=z), noise=noise)
    terms["loss"].mean().backward()
    out["kl/loss"] = terms["loss"].detach().numpy()
    out["kl/grad_sqsum"] = np.float64(sum((p.grad.double() ** 2).sum().item() for p in params if p.grad is not None))
    for nme in ("out.2.weight", "out.2.bias", "input_blocks.1.0.in_layers.2.weight"):
        flat_probe(f"kl/grad/{nme}", params[names.index(nme)].grad, out)

    # (e) calc_bpd_loop on an 8-step respaced chain (learned range), noise replayed from the seed
    model, diff, base = make("T28", respacing="8", learn_sigma=True)
    model.eval()
    th.manual_seed(9)
    bpd = diff.calc_bpd_loop(model, x0, clip_denoised=True, model_kwargs=dict(c=c, y=y, z=z))
    th.manual_seed(9)
    out["bpd/noise"] = th.stack([th.randn_like(x0) for _ in range(8)]).numpy()
    for k, v in bpd.items():
        out[f"bpd/{k}"] = v.numpy()
    np.savez_compressed(os.path.join(out_dir, "g9_vlb.npz"), **out)


# --------------------------------------------------------------------------- G10
def g10_flow(out_dir):
    """flow_based=True: MultivariateCausalFlow in place of the masked MLP layer (nn.py:342-426, unet.py:580-587)."""
    out = {}
    N = 4
    model, diff, base = make("T28", flow_based=True)
    out["keys"] = np.array(list(model.state_dict().keys()))
    out["shapes"] = np.array([str(tuple(v.shape)) for v in model.state_dict().values()])
    model.train()
    x0 = synth("G10.x0", (N, 1, 28, 28), 0.0, 1.0)
    c = synth("G10.c", (N, 2), 0.0, 1.0)
    y = th.tensor([0, 2, 4, 6], dtype=th.int64)
    t = th.tensor([3, 250, 600, 998], dtype=th.int64)
    noise = synth("G10.noise", (N, 1, 28, 28), -1.7, 1.7)
    # the flow alone
    mu = synth("G10.mu", (N, 512), -1.0, 1.0)
    C = th.eye(2) - th.tensor([[0, 1], [0, 0]], dtype=th.float32)
    with th.no_grad():
        z_post, log_det = model.causal_flow.flow(mu, C)
        rev_log_det, log_prob = model.causal_flow.reverse(z_post, C)
    out["flow/z_post"], out["flow/log_det"] = z_post.numpy(), log_det.numpy()
    out["flow/rev_log_det"], out["flow/log_prob"] = rev_log_det.numpy(), log_prob.numpy()
    # training_losses through the flow
    diff.kl_weight = 0.5
    params = list(model.parameters())
    names = [k for k, _ in model.named_parameters()]
    th.manual_seed(11)
    terms = diff.training_losses(model, x0, t, model_kwargs=dict(c=c, y=y), noise=noise, rep_cond=True, causal_modeling=True)
    th.manual_seed(11)
    out["eps_draw"] = th.randn(N, 512).numpy()
    terms["loss"].mean().backward()
    for k in ("loss", "mse", "kld_rep"):
        out[f"train/{k}"] = terms[k].detach().numpy()
    out["train/grad_sqsum"] = np.float64(sum((p.grad.double() ** 2).sum().item() for p in params if p.grad is not None))
    for nme in ("causal_flow.s_cond.0.weight", "causal_flow.s_cond.4.bias", "causal_flow.t_cond.2.weight", "rep_emb.fc_mu.weight", "out.2.weight"):
        flat_probe(f"train/grad/{nme}", params[names.index(nme)].grad, out)
    np.savez_compressed(os.path.join(out_dir, "g10_flow.npz"), **out)


# --------------------------------------------------------------------------- G11
def g11_variants(out_dir):
    """The two sibling model families of the evaluation scripts: label-conditional (context_cond, image_conditional_test.py:112-150)
    and DiffAE without the causal layer (image_diffae_test.py:262-322)."""
    out = {}
    N = 3
    x0 = synth("G11.x0", (N, 1, 28, 28), 0.0, 1.0)
    c = synth("G11.c", (N, 2), 0.0, 1.0)
    y = th.tensor([2, 4, 9], dtype=th.int64)
    noise = synth("G11.noise", (N, 1, 28, 28), -1.7, 1.7)
    tt = th.tensor([10, 400, 990], dtype=th.int64)
    # (a) label-conditional; the label vector is CONTEXT_DIM = 4 wide whatever n_vars says (script_util.py:10)
    c4 = synth("G11.c4", (N, 4), 0.0, 1.0)
    for phase in ("train", "sample"):
        model, diff, base = make("T28", respacing="" if phase == "train" else "ddim5", rep_cond=False, causal_modeling=False, context_cond=True)
        if phase == "train":
            out["cond/keys"] = np.array(list(model.state_dict().keys()))
            model.train()
            terms = diff.training_losses(model, x0, tt, model_kwargs=dict(c=c4, y=y), noise=noise)
            terms["loss"].mean().backward()
            out["cond/train/loss"], out["cond/train/mse"] = terms["loss"].detach().numpy(), terms["mse"].detach().numpy()
            params = dict(model.named_parameters())
            for nme in ("c_emb.0.weight", "c_emb.2.bias", "out.2.weight"):
                flat_probe(f"cond/train/grad/{nme}", params[nme].grad, out)
        else:
            model.eval()
            t_last = th.full((N,), diff.num_timesteps - 1, dtype=th.int64)
            x_t = diff.q_sample(x0, t_last, noise=noise)
            cc = c4.clone()
            cc[:, 0] = -0.2
            with th.no_grad():
                out["cond/sample"] = diff.ddim_sample_loop(model, (N, 1, 28, 28), noise=x_t, clip_denoised=True,
                                                           model_kwargs=dict(c=cc, y=y)).numpy()
    # (b) DiffAE: representation conditioning without the causal layer
    for phase in ("train", "sample"):
        model, diff, base = make("T28", respacing="" if phase == "train" else "ddim5", causal_modeling=False)
        if phase == "train":
            out["diffae/keys"] = np.array(list(model.state_dict().keys()))
            model.train()
            diff.kl_weight = 0.3
            th.manual_seed(21)
            terms = diff.training_losses(model, x0, tt, model_kwargs=dict(c=c, y=y), noise=noise, rep_cond=True, causal_modeling=False)
            th.manual_seed(21)
            out["diffae/eps_draw"] = th.randn(N, 512).numpy()
            terms["loss"].mean().backward()
            for k in ("loss", "mse", "kld_rep"):
                out[f"diffae/train/{k}"] = terms[k].detach().numpy()
            params = dict(model.named_parameters())
            for nme in ("rep_emb.fc_var.weight", "up_emb.weight", "out.2.weight"):
                flat_probe(f"diffae/train/grad/{nme}", params[nme].grad, out)
        else:
            model.eval()
            with th.no_grad():
                mu, var = model.rep_emb.encode(x0)
                var = th.ones(mu.shape) * 0.001
                mu[:, 256:512] = th.ones((N, 256)) * 0.4
                th.manual_seed(22)
                z = rnn.reparameterize(mu, var)
                th.manual_seed(22)
                out["diffae/z_eps"] = th.randn(N, 512).numpy()
                out["diffae/z"] = z.numpy()
                t_last = th.full((N,), diff.num_timesteps - 1, dtype=th.int64)
                x_t = diff.q_sample(x0, t_last, noise=noise)
                out["diffae/sample"] = diff.ddim_sample_loop(model, (N, 1, 28, 28), noise=x_t, clip_denoised=True,
                                                             model_kwargs=dict(c=c, y=y, z=z)).numpy()
    np.savez_compressed(os.path.join(out_dir, "g11_variants.npz"), **out)


# --------------------------------------------------------------------------- G12: one training step of the FULL models
def grad_probe(prefix, t, out):
    f = t.detach().double().flatten()
    out[f"{prefix}/head"] = f[:16].float().numpy()
    out[f"{prefix}/strided"] = f[::4999].float().numpy()
    out[f"{prefix}/absmax"] = np.float64(f.abs().max().item())
    out[f"{prefix}/sumsq"] = np.float64((f * f).sum().item())


def g12_full_train(out_dir):
    """training_losses + backward (gaussian_diffusion.py:768-859) on the benchmarked 41 M / 93 M parameter models at N = 2."""
    out = {}
    N = 2
    for tag in ["M32", "C64"]:
        model, diff, base = make(tag)
        model.train()
        x, x0, c, z, y = model_inputs(tag + ".train", base, N)
        t = th.tensor([37, 990], dtype=th.int64)
        noise = synth(tag + ".train.noise", tuple(x0.shape), -1.7, 1.7)
        diff.kl_weight = 0.3
        kw = dict(c=c)
        if y is not None:
            kw["y"] = y
        th.manual_seed(41)
        terms = diff.training_losses(model, x0, t, model_kwargs=kw, noise=noise, rep_cond=True, causal_modeling=True)
        th.manual_seed(41)
        out[f"{tag}/eps_draw"] = th.randn(N, 512).numpy()
        terms["loss"].mean().backward()
        for k in ("loss", "mse", "kld_rep"):
            out[f"{tag}/{k}"] = terms[k].detach().numpy()
        names = []
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            grad_probe(f"{tag}/g/{k}", p.grad, out)
        out[f"{tag}/grad_names"] = np.array(names)
        out[f"{tag}/grad_sqsum"] = np.float64(sum((p.grad.double() ** 2).sum().item() for p in model.parameters() if p.grad is not None))
        for k, v in model.state_dict().items():
            if "running_mean" in k or "running_var" in k:
                out[f"{tag}/after/{k}"] = v.numpy().copy()
    np.savez_compressed(os.path.join(out_dir, "g12_full_train.npz"), **out)


# --------------------------------------------------------------------------- G13: guidance w (two forwards per step)
class _RepDimZeros:
    """gaussian_diffusion.py:281 builds the unconditional z as zeros(N, 64), which only fits REP_DIM = 64 (the committed models have
    512: the call crashes).  The reference line runs unchanged with the ONE shape corrected to the model's rep_dim."""

    def __enter__(self):
        self.orig = th.zeros

        def zeros(*a, **k):
            if len(a) == 1 and isinstance(a[0], tuple) and len(a[0]) == 2 and a[0][1] == 64:
                return self.orig((a[0][0], 512), **k)
            return self.orig(*a, **k)
        th.zeros = zeros

    def __exit__(self, *exc):
        th.zeros = self.orig


def g13_guidance(out_dir):
    out = {}
    N = 2
    model, diff, base = make("P64", respacing="ddim100")
    model.eval()
    x, x0, c, z, _ = model_inputs("P64", base, N)
    with th.no_grad():
        noise = synth("P64.qnoise", (N, 4, 64, 64), -1.7, 1.7)
        x_t = diff.q_sample(x0, th.full((N,), 99, dtype=th.int64), noise=noise)
        for tv in (99, 40):
            tt = th.full((N,), tv, dtype=th.int64)
            for w in (0.5, 2.0):
                with _RepDimZeros():
                    o = diff.ddim_sample(model, x_t, tt, model_kwargs=dict(z=z), w=w)
                    pm = diff.p_mean_variance(model, x_t, tt, model_kwargs=dict(z=z), w=w)
                out[f"t{tv}/w{w}/sample"], out[f"t{tv}/w{w}/pred_xstart"] = o["sample"].numpy(), o["pred_xstart"].numpy()
                out[f"t{tv}/w{w}/mean"] = pm["mean"].numpy()
    np.savez_compressed(os.path.join(out_dir, "g13_guidance.npz"), **out)


# --------------------------------------------------------------------------- G14: p_sample_loop end to end
def g14_p_sample_loop(out_dir):
    """Ancestral sampling loop (gaussian_diffusion.py:416-504) on M32, respaced to 20 steps; the per-step noise the loop draws
    (th.randn_like, :402) is reproduced from the same seed and stored."""
    out = {}
    N = 2
    model, diff, base = make("M32", respacing="20")
    model.eval()
    x, x0, c, z, y = model_inputs("M32", base, N)
    with th.no_grad():
        x_T = synth("M32.xT", (N, 1, 32, 32), -1.7, 1.7)
        th.manual_seed(31)
        k = 0
        for o in diff.p_sample_loop_progressive(model, (N, 1, 32, 32), noise=x_T, model_kwargs=dict(z=z, y=y)):
            k += 1
            if k in (1, 10, 20):
                out[f"sample_after{k}"] = o["sample"].numpy()
        assert k == 20
        th.manual_seed(31)
        final = diff.p_sample_loop(model, (N, 1, 32, 32), noise=x_T, model_kwargs=dict(z=z, y=y))
        assert th.equal(final, th.from_numpy(out["sample_after20"]))
        th.manual_seed(31)
        out["step_noise"] = th.stack([th.randn(N, 1, 32, 32) for _ in range(20)]).numpy()
    np.savez_compressed(os.path.join(out_dir, "g14_p_sample_loop.npz"), **out)


# --------------------------------------------------------------------------- G15: BASELINE config [1] loss curve (fp32 reference)
def g15_m32_b256(out_dir):
    """Three optimizer steps of the reference on MorphoMNIST-shaped data at batch 256 (BASELINE config 'MorphoMNIST 32x32 CausalDiffAE
    training, bf16, batch 256'): the fp32 losses the reduced-precision torso is judged against (rel 2e-2)."""
    from torch.optim import AdamW
    out = {}
    N = 256
    model, diff, base = make("M32")
    model.train()
    params = list(model.parameters())
    opt = AdamW(params, lr=1e-4, weight_decay=0.0)
    diff.kl_weight = 0.1
    for step in range(3):
        x0 = synth(f"M32b.{step}.x0", (N, 1, 32, 32), 0.0, 1.0)
        c = synth(f"M32b.{step}.c", (N, 2), 0.0, 1.0)
        y = th.tensor([(step + 3 * i) % 10 for i in range(N)], dtype=th.int64)
        t = th.tensor([(137 * (step + 1) + 251 * i) % 1000 for i in range(N)], dtype=th.int64)
        noise = synth(f"M32b.{step}.noise", (N, 1, 32, 32), -1.7, 1.7)
        for p in params:
            p.grad = None
        th.manual_seed(200 + step)
        terms = diff.training_losses(model, x0, t, model_kwargs=dict(c=c, y=y), noise=noise, rep_cond=True, causal_modeling=True)
        th.manual_seed(200 + step)
        eps = th.randn(N, 512)
        out[f"step{step}/eps_draw_check"] = np.array([eps.double().sum().item(), (eps.double() ** 2).sum().item()] + eps.flatten()[:6].tolist())
        terms["loss"].mean().backward()
        for k in ("loss", "mse", "kld_rep"):
            out[f"step{step}/{k}_mean"] = np.float64(terms[k].detach().double().mean().item())
        out[f"step{step}/loss"] = terms["loss"].detach().numpy()
        opt.step()
        print("  G15 step", step, float(terms["loss"].mean()), flush=True)
    np.savez_compressed(os.path.join(out_dir, "g15_m32_b256.npz"), **out)


# --------------------------------------------------------------------------- G21: a 24-step fp32 loss curve the 16-bit torso is trained against
def g21_m32_curve(out_dir):
    """24 optimizer steps of the reference (fp32, AdamW lr 1e-4, kl_weight 0.1) on the M32 model at batch 16, data cycling through a pool of
    four closed-form batches: the loss / mse / kld_rep means per step.  tests/test_gpu_torso16.py trains the SAME run through TrainLoop in the
    parity mode (tight) and on the 16-bit torso (use_fp16: 2e-2) and then carries both on to 120 steps (train_util.py:231-297)."""
    from torch.optim import AdamW
    out = {}
    N, STEPS = 16, 24
    model, diff, base = make("M32")
    model.train()
    params = list(model.parameters())
    opt = AdamW(params, lr=1e-4, weight_decay=0.0)
    diff.kl_weight = 0.1
    for step in range(STEPS):
        b = step % 4
        x0 = synth(f"M32c.{b}.x0", (N, 1, 32, 32), -1.0, 1.0)
        c = synth(f"M32c.{b}.c", (N, 2), 0.0, 1.0)
        y = th.tensor([(b + 3 * i) % 10 for i in range(N)], dtype=th.int64)
        t = th.tensor([(137 * (step + 1) + 251 * i) % 1000 for i in range(N)], dtype=th.int64)
        noise = synth_noise(f"M32c.{step}.noise", (N, 1, 32, 32))
        for p in params:
            p.grad = None
        th.manual_seed(300 + step)
        terms = diff.training_losses(model, x0, t, model_kwargs=dict(c=c, y=y), noise=noise, rep_cond=True, causal_modeling=True)
        th.manual_seed(300 + step)
        eps = th.randn(N, 512)
        out[f"step{step}/eps_draw_check"] = np.array([eps.double().sum().item(), (eps.double() ** 2).sum().item()] + eps.flatten()[:6].tolist())
        terms["loss"].mean().backward()
        for k in ("loss", "mse", "kld_rep"):
            out[f"step{step}/{k}_mean"] = np.float64(terms[k].detach().double().mean().item())
        opt.step()
        print("  G21 step", step, float(terms["loss"].mean()), float(terms["mse"].mean()), flush=True)
    out["steps"], out["batch"] = np.int64(STEPS), np.int64(N)
    np.savez_compressed(os.path.join(out_dir, "g21_m32_curve.npz"), **out)


# --------------------------------------------------------------------------- G22: resampling without a conv (conv_resample=False)
def g22_plain_resample(out_dir):
    """runet.Upsample(channels, use_conv=False) (unet.py:51-78: F.interpolate alone): output and input gradient on a closed-form tensor.
    The matching Downsample(use_conv=False) cannot be run: the reference builds it as avg_pool_nd(stride) — `dims` is missing, so
    nn.AvgPool2d() raises TypeError at construction (unet.py:101, nn.py:497) and UNetModel(conv_resample=False) never gets built; recorded
    here so that the fixture says why the pool is pinned to torch's avg_pool2d only."""
    out = {}
    up = runet.Upsample(64, False)
    x = synth("T28r.x", (2, 64, 7, 14)).requires_grad_(True)
    y = up(x)
    gy = synth("T28r.gy", tuple(y.shape))
    (y * gy).sum().backward()
    out["up/y"], out["up/dx"] = y.detach().numpy(), x.grad.numpy()
    try:
        runet.Downsample(64, False)
        out["down/reference_builds"] = np.int64(1)
    except TypeError as e:
        out["down/reference_builds"] = np.int64(0)
        out["down/error"] = np.array(str(e))
    np.savez_compressed(os.path.join(out_dir, "g22_plain_resample.npz"), **out)


# --------------------------------------------------------------------------- G16
def g16_dropout(out_dir):
    """Training-mode ResBlock with dropout > 0 (unet.py:153): the keep mask the reference's nn.Dropout drew is recorded next to the
    outputs (a forward hook on the module: kept <=> output != 0 wherever the input != 0), so the product can be fed the same mask."""
    out, meta = {}, {}
    for tag, ci, co, ssn, pdrop, shape in [("drop_same", 128, 128, True, 0.3, (2, 128, 8, 8)), ("drop_skip", 64, 96, False, 0.1, (3, 64, 6, 6))]:
        blk = runet.ResBlock(ci, 512, pdrop, out_channels=co, use_scale_shift_norm=ssn)
        load_closed_form(blk, tag + ".")
        blk.train()
        seen = {}
        drop = [m for m in blk.out_layers if isinstance(m, th.nn.Dropout)][0]
        drop.register_forward_hook(lambda m, i, o: seen.update(mask=((o != 0) | (i[0] == 0)).float().detach()))
        th.manual_seed(1234)
        x = synth(tag + ".x", shape).requires_grad_(True)
        e = synth(tag + ".emb", (shape[0], 512)).requires_grad_(True)
        y = blk(x, e)
        gy = synth(tag + ".gy", tuple(y.shape))
        (y * gy).sum().backward()
        out[f"{tag}/mask"] = seen["mask"].numpy().astype(np.uint8)
        kept = float(seen["mask"].mean())
        out[f"{tag}/y"], out[f"{tag}/gx"], out[f"{tag}/gemb"] = y.detach().numpy(), x.grad.numpy(), e.grad.numpy()
        for k, p in blk.named_parameters():
            flat_probe(f"{tag}/g.{k}", p.grad, out)
        blk.eval()
        with th.no_grad():
            out[f"{tag}/y_eval"] = blk(x, e).numpy()
        meta[tag] = dict(ci=ci, co=co, ssn=ssn, p=pdrop, x_shape=list(shape), y_shape=list(y.shape), kept=kept)
    np.savez_compressed(os.path.join(out_dir, "g16_dropout.npz"), **out)
    with open(os.path.join(out_dir, "g16_dropout.json"), "w") as f:
        json.dump(meta, f, indent=1)



# --------------------------------------------------------------------------- G17..G20: the long loops and a trained-like weight distribution
def _counterfactual_start(model, diff, base, N, t_last, A, tag="P64"):
    """encode -> causal layer -> do(z_post[:, :128] := 0.2) -> reparameterize(var = 0.001) -> q_sample(t_last): the pendulum branch of
    image_causaldae_test.py:535-594; returns (x_t, z, eps draw)"""
    x, x0, c, z, _ = model_inputs(tag, base, N)
    mu, var = model.rep_emb.encode(x0)
    var = th.ones_like(mu) * 0.001
    z_pre = model.causal_mask.causal_masking(mu, A)
    z_post = model.causal_mask.nonlinearity_add_back_noise(mu, z_pre)
    z_post[:, :128] = 0.2
    th.manual_seed(5)
    zz = rnn.reparameterize(z_post, var)
    th.manual_seed(5)
    eps = th.randn(N, 512)
    noise = synth(tag + ".qnoise", (N, base["in_channels"], base["image_size"], base["image_size"]), -1.7, 1.7)
    x_t = diff.q_sample(x0, th.full((N,), t_last, dtype=th.int64), noise=noise)
    return x_t, zz, eps


def g17_ddim250(out_dir):
    """BASELINE config [4]: P64, "ddim250" (respace.py:30-37), the whole 250-step deterministic loop (gaussian_diffusion.py:598-680)."""
    out = {}
    N = 2
    model, diff, base = make("P64", respacing="ddim250")
    model.eval()
    assert diff.num_timesteps == 250
    A = th.tensor(ADJ["pendulum"], dtype=th.float32)
    with th.no_grad():
        x_t, zz, eps = _counterfactual_start(model, diff, base, N, 249, A)
        out["x_t"], out["z"], out["eps_draw"] = x_t.numpy(), zz.numpy(), eps.numpy()
        k = 0
        for o in diff.ddim_sample_loop_progressive(model, (N, 4, 64, 64), noise=x_t, model_kwargs=dict(z=zz)):
            k += 1
            if k in (1, 25, 125, 250):
                out[f"sample_after{k}"] = o["sample"].numpy()
        assert k == 250
    np.savez_compressed(os.path.join(out_dir, "g17_ddim250.npz"), **out)


class _ClosedFormRandnLike:
    """th.randn_like -> a closed-form unit-variance draw by call count (oracle.closed_form.synth_noise), so that a 1000-step ancestral
    loop (one randn_like per step, gaussian_diffusion.py:402) is reproducible without storing 1000 noise tensors."""

    def __init__(self, prefix):
        self.prefix, self.k = prefix, 0

    def __enter__(self):
        self.orig = th.randn_like

        def fake(x, **kw):
            v = synth_noise(f"{self.prefix}.{self.k}", tuple(x.shape))
            self.k += 1
            return v

        th.randn_like = fake
        return self

    def __exit__(self, *exc):
        th.randn_like = self.orig
        return False


def g18_p_sample_t1000(out_dir):
    """BASELINE config [0]: M32, T = 1000, ancestral sampling end to end (gaussian_diffusion.py:416-504), batch 2."""
    out = {}
    N = 2
    model, diff, base = make("M32", respacing="")
    model.eval()
    assert diff.num_timesteps == 1000
    x, x0, c, z, y = model_inputs("M32", base, N)
    with th.no_grad():
        x_T = synth("M32.xT", (N, 1, 32, 32), -1.7, 1.7)
        with _ClosedFormRandnLike("G18.noise") as rl:
            k = 0
            for o in diff.p_sample_loop_progressive(model, (N, 1, 32, 32), noise=x_T, model_kwargs=dict(z=z, y=y)):
                k += 1
                if k in (1, 10, 100, 500, 900, 1000):
                    out[f"sample_after{k}"] = o["sample"].numpy()
                    out[f"pred_xstart_after{k}"] = o["pred_xstart"].numpy()
            assert k == 1000 and rl.k == 1000
    np.savez_compressed(os.path.join(out_dir, "g18_p_sample_t1000.npz"), **out)


def load_trained_like(module):
    sd = module.state_dict()
    for k in sd:
        sd[k] = fill_value_trained(k, sd[k].shape)
    module.load_state_dict(sd)
    return module


def g19_trained_like(out_dir):
    """P64 with a TRAINED-LIKE weight distribution (oracle.closed_form.fill_value_trained: log-uniform magnitudes over four decades,
    a quarter of the zero-initialised layers left zero): one forward, the encoder path, and the DDIM-100 loop."""
    out = {}
    N = 2
    model, diff, base = make("P64", respacing="ddim100")
    load_trained_like(model)
    model.eval()
    A = th.tensor(ADJ["pendulum"], dtype=th.float32)
    x, x0, c, z, _ = model_inputs("P64", base, N)
    zero = [k for k, v in model.state_dict().items() if v.dim() >= 2 and float(v.abs().max()) == 0.0]
    out["zero_keys"] = np.array(zero)
    with th.no_grad():
        t = th.tensor([37.0, 990.0])
        e, *_ = model(x, t, z=z)
        out["eps_z"] = e.numpy()
        x_t, zz, eps = _counterfactual_start(model, diff, base, N, 99, A)
        out["x_t"], out["z"], out["eps_draw"] = x_t.numpy(), zz.numpy(), eps.numpy()
        o = diff.ddim_sample(model, x_t, th.full((N,), 99, dtype=th.int64), model_kwargs=dict(z=zz))
        out["step99/sample"], out["step99/pred_xstart"] = o["sample"].numpy(), o["pred_xstart"].numpy()
        k = 0
        for o in diff.ddim_sample_loop_progressive(model, (N, 4, 64, 64), noise=x_t, model_kwargs=dict(z=zz)):
            k += 1
            if k in (1, 10, 50, 100):
                out[f"loop/sample_after{k}"] = o["sample"].numpy()
    np.savez_compressed(os.path.join(out_dir, "g19_trained_like.npz"), **out)


def g20_traversal(out_dir):
    """The latent traversal of the evaluation script (image_causaldae_test.py:481-531, pendulum branch, `traversal == True`), restated
    call for call on the reference's model and diffusion: x_t = q_sample(batch, t = 249, noise) ONCE; then for eight values
    -0.5, -0.35, ... (value += 0.15 in a Python float): mu[:, 16:32] = value (the script's hard-coded columns, before the causal layer),
    z = reparameterize(causal layer(mu), 0.001) with a FRESH draw per value, ddim_sample_loop(noise = x_t, z).  "ddim250" so that
    t = 249 is the last spaced step, as in the script's launch configuration."""
    out = {}
    N = 2
    model, diff, base = make("P64", respacing="ddim250")
    model.eval()
    A = th.tensor(ADJ["pendulum"], dtype=th.float32)
    x, x0, c, z, _ = model_inputs("P64", base, N)
    batch = x0
    with th.no_grad():
        noise = synth("P64.qnoise", (N, 4, 64, 64), -1.7, 1.7)          # the script: th.randn_like(batch)
        t = th.ones((batch.shape[0]), dtype=th.int64) * 249
        x_t = diff.q_sample(batch, t, noise=noise)
        out["x_t"] = x_t.numpy()
        value = -0.5
        values, draws = [], []
        for i in range(8):
            mu, var = model.rep_emb.encode(batch)
            var = th.ones(mu.shape) * 0.001
            mu[:, 16:32] = th.ones((N, 16)) * value
            z_pre = model.causal_mask.causal_masking(mu, A)
            z_post = model.causal_mask.nonlinearity_add_back_noise(mu, z_pre)
            th.manual_seed(100 + i)
            zz = rnn.reparameterize(z_post, var)
            th.manual_seed(100 + i)
            draws.append(th.randn(N, 512).numpy())
            sample = diff.ddim_sample_loop(model, (N, 4, 64, 64), noise=x_t, clip_denoised=True, model_kwargs=dict(z=zz), w=None)
            out[f"z{i}"], out[f"sample{i}"] = zz.numpy(), sample.numpy()
            values.append(value)
            value += 0.15
        out["values"] = np.array(values, dtype=np.float64)
        out["eps_draws"] = np.stack(draws)
    np.savez_compressed(os.path.join(out_dir, "g20_traversal.npz"), **out)


ALL = dict(G16=g16_dropout, G1=g1_schedules, G2=g2_temb, G3=g3_blocks, G4=g4_encoder, G5=g5_rep_loss, G6=g6_unet, G7=g7_train, G8=g8_ddim, G9=g9_vlb, G10=g10_flow, G11=g11_variants, G12=g12_full_train, G13=g13_guidance, G14=g14_p_sample_loop, G15=g15_m32_b256, G17=g17_ddim250, G18=g18_p_sample_t1000, G19=g19_trained_like, G20=g20_traversal, G21=g21_m32_curve, G22=g22_plain_resample)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    th.set_num_threads(os.cpu_count())
    for name, fn in ALL.items():
        if a.only and name not in a.only.split(","):
            continue
        print("generating", name, flush=True)
        fn(a.out)
