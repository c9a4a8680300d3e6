"""Pin the oracle (oracle/*.py) against golden vectors produced by the reference itself
(tools/gen_golden.py, run in the build container).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_ref as D
from oracle import schedule as S
from oracle import unet_ref as U
from oracle.closed_form import fill_state_dict, fill_state_dict_trained, fill_value, synth, synth_noise

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-6          # same ATen CPU kernels, different op grouping


def close(a, b, tol=TOL, rel=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    bound = tol + rel * np.abs(b).max() if b.size else tol
    assert err <= bound, f"max|d|={err:.3e} > {bound:.3e}"


def probe_close(t, g, prefix, tol=TOL, rel=1e-5):
    f = t.detach().double().flatten()
    close(f[:64].numpy(), g[prefix + "/head"], tol, rel)
    close(f[::997].numpy(), g[prefix + "/strided"], tol, rel)
    s, ss = float(g[prefix + "/sum"]), float(g[prefix + "/sumsq"])
    assert abs(f.sum().item() - s) <= 1e-4 * max(1.0, abs(s)) + 1e-3 * np.sqrt(ss)
    assert abs((f * f).sum().item() - ss) <= 1e-4 * max(ss, 1e-12)


# ------------------------------------------------------------------ G1: schedules (f64 tables bit-exact, int sets exact)
@pytest.mark.parametrize("rs", ["", "ddim100", "ddim250", "250", "100", "ddim50", "10,10,10"])
def test_schedule_tables(golden, rs):
    g = golden("g1_schedules.npz")
    tab, tmap, T0 = S.make_schedule(1000, "linear", rs)
    tag = rs or "full"
    assert T0 == 1000
    np.testing.assert_array_equal(np.array(tmap, dtype=np.int64), g[f"{tag}/timestep_map"])
    for n in S.TABLE_NAMES + ("fixed_large_variance",):
        np.testing.assert_array_equal(tab[n], g[f"{tag}/{n}"], err_msg=n)     # float64, bit exact


def test_cosine_and_space_timesteps(golden):
    g = golden("g1_schedules.npz")
    np.testing.assert_array_equal(S.make_schedule(500, "cosine", "")[0]["betas"], g["cosine500/betas"])
    for key in [k for k in g.files if k.startswith("space/") and k != "space/errors"]:
        _, T, spec = key.split("/")
        got = S.space_timesteps(int(T), spec if spec.startswith("ddim") or "," not in spec else [int(s) for s in spec.split(",")])
        np.testing.assert_array_equal(np.array(sorted(got), dtype=np.int64), g[key], err_msg=key)
    errs = []
    for T, spec in [(1000, "ddim7"), (10, "20"), (1000, "ddim999")]:
        try:
            S.space_timesteps(T, spec)
            errs.append(0)
        except ValueError:
            errs.append(1)
    np.testing.assert_array_equal(np.array(errs), g["space/errors"])
    assert S.space_timesteps(1000, "ddim250") != S.space_timesteps(1000, "250")


# ------------------------------------------------------------------ G2
def test_timestep_embedding_and_wrapped_t(golden):
    g = golden("g2_temb.npz")
    t = torch.from_numpy(g["t"])
    close(U.timestep_embedding(t, 128).numpy(), g["temb128"], 1e-6)
    close(U.timestep_embedding(t, 33).numpy(), g["temb33"], 1e-6)
    sch = D.Schedule(1000, "linear", "ddim100", True)
    out = sch.model_t(torch.from_numpy(g["wrapped_in"]))
    np.testing.assert_array_equal(out.numpy(), g["wrapped_out"])
    assert out.dtype == torch.float32
    sch2 = D.Schedule(1000, "linear", "250", False)
    out2 = sch2.model_t(torch.tensor([0, 248, 249]))
    np.testing.assert_array_equal(out2.numpy(), g["wrapped_norescale_out"])
    assert out2.dtype == torch.int64


# ------------------------------------------------------------------ G3: blocks fwd + grads
def _block_sd(tag, spec):
    sd = {k: fill_value(f"{tag}.{k}", s) for k, s in spec}
    return {k: v.requires_grad_(True) for k, v in sd.items()}


def _strip(spec, prefix):
    return [(k[len(prefix) + 1:], s) for k, s in spec]


CASES_RES = [("res_same", 128, 128, True), ("res_skip", 128, 256, True), ("res_cat", 384, 128, True), ("res_nossn", 64, 96, False)]


@pytest.mark.parametrize("tag,ci,co,ssn", CASES_RES)
def test_resblock(golden, tag, ci, co, ssn):
    g = golden("g3_blocks.npz")
    meta = json.load(open(os.path.join(GOLDEN, "g3_blocks.json")))[tag]
    spec = _strip(U._layer_spec("b", ("res", ci, co), 512, ssn), "b")
    sd = _block_sd(tag, spec)
    x = synth(tag + ".x", meta["x_shape"]).requires_grad_(True)
    e = synth(tag + ".emb", (meta["x_shape"][0], 512)).requires_grad_(True)
    y = U.resblock({"b." + k: v for k, v in sd.items()}, "b", x, e, ssn)
    (y * synth(tag + ".gy", meta["y_shape"])).sum().backward()
    close(y.detach().numpy(), g[tag + "/y"], 1e-5)
    close(x.grad.numpy(), g[tag + "/gx"], 1e-5, 1e-5)
    close(e.grad.numpy(), g[tag + "/gemb"], 1e-5, 1e-5)
    for k, v in sd.items():
        probe_close(v.grad, g, f"{tag}/g.{k}", 1e-5)


@pytest.mark.parametrize("tag", ["drop_same", "drop_skip"])
def test_resblock_training_dropout(golden, tag):
    """G16: the reference block in training mode with dropout > 0 (unet.py:153); the oracle takes the mask the reference drew."""
    g = golden("g16_dropout.npz")
    meta = json.load(open(os.path.join(GOLDEN, "g16_dropout.json")))[tag]
    ci, co, ssn, pd = meta["ci"], meta["co"], meta["ssn"], meta["p"]
    spec = _strip(U._layer_spec("b", ("res", ci, co), 512, ssn), "b")
    sd = _block_sd(tag, spec)
    x = synth(tag + ".x", meta["x_shape"]).requires_grad_(True)
    e = synth(tag + ".emb", (meta["x_shape"][0], 512)).requires_grad_(True)
    drop = torch.from_numpy(g[tag + "/mask"]).float() / (1.0 - pd)
    bsd = {"b." + k: v for k, v in sd.items()}
    y = U.resblock(bsd, "b", x, e, ssn, drop=drop)
    (y * synth(tag + ".gy", meta["y_shape"])).sum().backward()
    close(y.detach().numpy(), g[tag + "/y"], 1e-5)
    close(x.grad.numpy(), g[tag + "/gx"], 1e-5, 1e-5)
    close(e.grad.numpy(), g[tag + "/gemb"], 1e-5, 1e-5)
    for k, v in sd.items():
        probe_close(v.grad, g, f"{tag}/g.{k}", 1e-5)
    with torch.no_grad():
        close(U.resblock(bsd, "b", x, e, ssn).numpy(), g[tag + "/y_eval"], 1e-5)


@pytest.mark.parametrize("ch,T", [(96, 256), (128, 64), (64, 256), (64, 16)])
def test_attention_block(golden, ch, T):
    g = golden("g3_blocks.npz")
    tag = f"attn_{ch}_{T}"
    meta = json.load(open(os.path.join(GOLDEN, "g3_blocks.json")))[tag]
    spec = _strip(U._layer_spec("b", ("attn", ch * 4, 4), 512, True), "b")
    sd = _block_sd(tag, spec)
    x = synth(tag + ".x", meta["x_shape"]).requires_grad_(True)
    y = U.attnblock({"b." + k: v for k, v in sd.items()}, "b", x, 4)
    (y * synth(tag + ".gy", meta["y_shape"])).sum().backward()
    close(y.detach().numpy(), g[tag + "/y"], 1e-5)
    close(x.grad.numpy(), g[tag + "/gx"], 1e-5, 1e-5)
    for k, v in sd.items():
        probe_close(v.grad, g, f"{tag}/g.{k}", 1e-5)


def test_qkv_attention(golden):
    g = golden("g3_blocks.npz")
    qkv = synth("qkv.x", (8, 96, 64), -2, 2).requires_grad_(True)
    # oracle takes [B, 3C, T] + heads; the golden tensor is already [B*H, 3ch, T] => heads=1
    y = U.qkv_attention(qkv, 1)
    (y * synth("qkv.gy", tuple(y.shape))).sum().backward()
    close(y.detach().numpy(), g["qkv/y"], 1e-6)
    close(qkv.grad.numpy(), g["qkv/gx"], 1e-6)


@pytest.mark.parametrize("tag,kind,xs", [("down", "down", (2, 128, 8, 8)), ("up", "up", (2, 128, 4, 4))])
def test_resample(golden, tag, kind, xs):
    g = golden("g3_blocks.npz")
    spec = _strip(U._layer_spec("b", (kind, 128), 512, True), "b")
    sd = _block_sd(tag, spec)
    x = synth(tag + ".x", xs).requires_grad_(True)
    y = U.run_layer({"b." + k: v for k, v in sd.items()}, "b", (kind, 128), x, None)
    (y * synth(tag + ".gy", tuple(y.shape))).sum().backward()
    close(y.detach().numpy(), g[tag + "/y"], 1e-5)
    close(x.grad.numpy(), g[tag + "/gx"], 1e-5)
    for k, v in sd.items():
        probe_close(v.grad, g, f"{tag}/g.{k}", 1e-5)


def test_head(golden):
    import torch.nn.functional as F
    g = golden("g3_blocks.npz")
    sd = _block_sd("head", [("0.weight", (128,)), ("0.bias", (128,)), ("2.weight", (4, 128, 3, 3)), ("2.bias", (4,))])
    x = synth("head.x", (2, 128, 8, 8)).requires_grad_(True)
    y = F.conv2d(U.silu(U.gn32(x, sd["0.weight"], sd["0.bias"])), sd["2.weight"], sd["2.bias"], padding=1)
    (y * synth("head.gy", tuple(y.shape))).sum().backward()
    close(y.detach().numpy(), g["head/y"], 1e-5)
    close(x.grad.numpy(), g["head/gx"], 1e-5)


# ------------------------------------------------------------------ G4: encoder + causal layer
@pytest.mark.parametrize("tag,C,S_,nv", [("enc32", 1, 32, 2), ("enc64", 4, 64, 4), ("enc96", 4, 96, 4)])
def test_encoder(golden, tag, C, S_, nv):
    g = golden("g4_encoder.npz")
    cfg = U.default_cfg(image_size=64 if S_ == 96 else S_, in_channels=C, n_vars=nv, rep_cond=True,
                        encoder_dims=U.encoder_dims(S_, nv))
    spec = [(k, s) for k, s in U.param_spec(cfg) if k.startswith("rep_emb.")]
    sd = fill_state_dict(spec)
    L = U.n_encoder_layers(sd)
    x = synth(str(g[tag + "/input_name"]), (4, C, S_, S_), 0.0, 1.0)       # the generator picks an input away from every LeakyReLU kink
    mu, var = U.encode(sd, x, L, training=False)
    close(mu.numpy(), g[tag + "/eval_mu"], 1e-5)
    close(var.numpy(), g[tag + "/eval_var"], 1e-5)
    for v in sd.values():
        if v.dtype == torch.float32:
            v.requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    new = {}
    mu, var = U.encode(sd, xg, L, training=True, new_stats=new)
    close(mu.detach().numpy(), g[tag + "/train_mu"], 2e-5)
    close(var.detach().numpy(), g[tag + "/train_var"], 2e-5)
    ((mu * synth(tag + ".gmu", (4, 512))).sum() + (var * synth(tag + ".gvar", (4, 512))).sum()).backward()
    close(xg.grad.numpy(), g[tag + "/train_gx"], 1e-5, 1e-4)
    for k, v in new.items():
        close(v.detach().numpy(), g[f"{tag}/after.{k[len('rep_emb.'):]}"], 1e-6)
    for k, v in sd.items():
        kk = f"{tag}/g.{k[len('rep_emb.'):]}/head"
        if k.endswith(".0.bias"):
            continue        # conv bias ahead of batch-stat BN: gradient is exactly 0 in theory, rounding noise in fp32
        if kk in g.files:
            probe_close(v.grad, g, kk[:-5], 2e-5, 1e-4)


def test_causal_layer(golden):
    g = golden("g4_encoder.npz")
    for nv, graphs in [(2, ["morpho"]), (4, ["circuit", "pendulum"])]:
        cfg = U.default_cfg(n_vars=nv, rep_cond=True, causal_modeling=True)
        spec = [(k, s) for k, s in U.param_spec(cfg) if k.startswith("causal_mask.")]
        sd = fill_state_dict(spec)
        u = synth(f"causal{nv}.u", (3, 512)).requires_grad_(True)
        for gname in graphs:
            A = torch.tensor(U.ADJ[gname], dtype=torch.float32)
            z_pre = U.causal_masking(u, A, nv)
            z_post = U.nonlinearity_add_back_noise(sd, u, z_pre, nv)
            close(z_pre.detach().numpy(), g[f"causal/{gname}/z_pre"], 1e-6)
            close(z_post.detach().numpy(), g[f"causal/{gname}/z_post"], 1e-5)
        (z_post * synth(f"causal{nv}.gz", (3, 512))).sum().backward()
        close(u.grad.numpy(), g[f"causal/{graphs[-1]}/gu"], 1e-5)
    m, v = synth("rep.m", (3, 512)), synth("rep.v", (3, 512), 0.01, 1.0)
    close(U.reparameterize(m, v, torch.from_numpy(g["reparam/eps"])).numpy(), g["reparam/z"], 1e-6)


# ------------------------------------------------------------------ G5
def test_representation_loss(golden):
    g = golden("g5_rep_loss.npz")
    for nv in (2, 4):
        N = 5
        mu, var = synth(f"rl{nv}.mu", (N, 512)), synth(f"rl{nv}.var", (N, 512), 0.05, 2.0)
        zp, c = synth(f"rl{nv}.zp", (N, 512)), synth(f"rl{nv}.c", (N, nv), 0.0, 1.0)
        mask = torch.tensor([1.0, 0.0, 1.0, 1.0, 0.0])
        close(D.representation_loss(mu, var, zp, True, None, c).numpy(), g[f"nv{nv}/causal"], 1e-3, 1e-6)
        close(D.representation_loss(mu, var, zp, False, None, c).numpy(), g[f"nv{nv}/plain"], 1e-3, 1e-6)
        close(D.representation_loss(mu, var, zp, True, mask, c).numpy(), g[f"nv{nv}/causal_masked"], 1e-3, 1e-6)


# ------------------------------------------------------------------ G6: full UNet forward (M32 / P64 / C64)
MODEL_CFG = {
    "M32": dict(image_size=32, in_channels=1, n_vars=2, class_cond=True),
    "P64": dict(image_size=64, in_channels=4, n_vars=4),
    "C64": dict(image_size=64, in_channels=3, n_vars=4),
    "T28": dict(image_size=28, in_channels=1, n_vars=2, class_cond=True, num_channels=32, num_res_blocks=1),
}


def model_cfg(tag, **over):
    return U.default_cfg(rep_cond=True, causal_modeling=True, **MODEL_CFG[tag], **over)


def model_inputs(tag, cfg, N):
    C, S_, nv = cfg["in_channels"], cfg["image_size"], cfg["n_vars"]
    x = synth(tag + ".x", (N, C, S_, S_))
    x0 = synth(tag + ".x0", (N, C, S_, S_), 0.0, 1.0)
    c = synth(tag + ".c", (N, nv), 0.0, 1.0)
    z = synth(tag + ".z", (N, 512))
    y = torch.tensor([(3 * i + 1) % 10 for i in range(N)], dtype=torch.int64) if cfg["class_cond"] else None
    return x, x0, c, z, y


@pytest.mark.parametrize("tag", ["M32", "P64", "C64"])
def test_unet_forward(golden, tag):
    g = golden("g6_unet.npz")
    cfg = model_cfg(tag)
    spec = U.param_spec(cfg)
    assert [k for k, _ in spec] == list(g[f"{tag}/keys"])                       # state-dict layout == reference
    assert [str(tuple(s)) for _, s in spec] == list(g[f"{tag}/shapes"])
    sd = fill_state_dict(spec)
    n_params = sum(int(np.prod(s)) for k, s in spec if "running" not in k and "num_batches" not in k)
    assert n_params == int(g[f"{tag}/n_params"])
    x, x0, c, z, y = model_inputs(tag, cfg, 2)
    t = torch.tensor([37.0, 990.0])
    with torch.no_grad():
        e, *_ = U.unet_forward(sd, cfg, x, t, y=y, z=z)
        close(e.numpy(), g[f"{tag}/eps_z"], 2e-5)
        e2, mu, var, zp, _ = U.unet_forward(sd, cfg, x, t, y=y, x_start=x0, eps_z=torch.from_numpy(g[f"{tag}/eps_draw"]))
        close(mu.numpy(), g[f"{tag}/mu"], 1e-5)
        close(var.numpy(), g[f"{tag}/var"], 1e-5)
        close(zp.numpy(), g[f"{tag}/z_post"], 1e-5)
        close(e2.numpy(), g[f"{tag}/eps_enc"], 2e-5)


# ------------------------------------------------------------------ G7: training_losses + AdamW/EMA trajectory
@pytest.mark.parametrize("variant,masking", [("plain", False), ("masked", True)])
def test_training_trajectory(golden, variant, masking):
    g = golden("g7_train.npz")
    cfg = model_cfg("T28", masking=masking)
    spec = U.param_spec(cfg)
    sd = fill_state_dict(spec)
    pkeys = [k for k, _ in spec if "running" not in k and "num_batches" not in k]
    params = [sd[k].requires_grad_(True) for k in pkeys]
    ema = [p.detach().clone() for p in params]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    sch = D.Schedule(1000, "linear", "", True)
    sel = list(g[f"{variant}/sel_names"])
    N = 4
    for step in range(3):
        x0 = synth(f"T28.{step}.x0", (N, 1, 28, 28), 0.0, 1.0)
        c = synth(f"T28.{step}.c", (N, 2), 0.0, 1.0)
        y = torch.tensor([(step + 2 * i) % 10 for i in range(N)], dtype=torch.int64)
        t = torch.from_numpy(g[f"{variant}/step{step}/t"])
        noise = synth(f"T28.{step}.noise", (N, 1, 28, 28), -1.7, 1.7)
        eps_z = torch.from_numpy(g[f"{variant}/step{step}/eps_draw"])
        mask = torch.from_numpy(g[f"{variant}/step{step}/cfg_mask"]) if masking else None
        new = {}

        def model_full(x_t, tm, xs):
            return U.unet_forward(sd, cfg, x_t, tm, y=y, c=c, x_start=xs, eps_z=eps_z, cfg_mask=mask, training=True, new_stats=new)

        for p in params:
            p.grad = None
        terms = D.training_losses(sch, model_full, x0, t, noise, c=c, rep_cond=True, causal_modeling=True,
                                  kl_weight=[0.0, 0.25, 0.5][step])
        terms["loss"].mean().backward()
        for k in ("loss", "mse", "kld_rep"):
            close(terms[k].detach().numpy(), g[f"{variant}/step{step}/{k}"], 1e-4, 1e-5)
        sq = sum((p.grad.double() ** 2).sum().item() for p in params)
        assert abs(sq - float(g[f"{variant}/step{step}/grad_sqsum"])) <= 1e-3 * sq
        if step == 0:
            for k in sel:
                probe_close(sd[k].grad, g, f"{variant}/grad0/{k}", 1e-5, 1e-3)
        with torch.no_grad():
            D.adamw_ema_step(params, [p.grad for p in params], m, v, ema, step + 1)
            for k, val in new.items():
                sd[k] = val
        if step in (0, 2):
            for k in sel:
                i = pkeys.index(k)
                probe_close(params[i], g, f"{variant}/after{step + 1}/{k}", 1e-5, 1e-4)
                probe_close(ema[i], g, f"{variant}/ema{step + 1}/{k}", 1e-6, 1e-5)
            close(sd["rep_emb.encoder.1.1.running_var"].numpy(), g[f"{variant}/after{step + 1}/bn_running_var"], 1e-6)
    close(np.array([D.kl_weight_at(s) for s in (0, 1, 2, 25000, 49999, 50000, 60000)]), g["kl_weight_sched"], 0.0)


# ------------------------------------------------------------------ G8: counterfactual pattern, single steps, DDIM-100 loop
def test_ddim_p64(golden):
    g = golden("g8_ddim.npz")
    cfg = model_cfg("P64")
    sd = fill_state_dict(U.param_spec(cfg))
    N = 2
    x, x0, c, z, _ = model_inputs("P64", cfg, N)
    sch = D.Schedule(1000, "linear", "ddim100", True)
    with torch.no_grad():
        A = torch.tensor(U.ADJ["pendulum"], dtype=torch.float32)
        mu, _ = U.encode(sd, x0, U.n_encoder_layers(sd))
        z_post = U.nonlinearity_add_back_noise(sd, mu, U.causal_masking(mu, A, 4), 4)
        z_post[:, :128] = 0.2
        zz = U.reparameterize(z_post, torch.full_like(mu, 0.001), torch.from_numpy(g["cf/eps_draw"]))
        close(zz.numpy(), g["cf/z"], 1e-5)
        t99 = torch.full((N,), 99, dtype=torch.int64)
        x_t = D.q_sample(sch, x0, t99, synth("P64.qnoise", (N, 4, 64, 64), -1.7, 1.7))
        close(x_t.numpy(), g["cf/x_t"], 1e-6)
        model_fn = lambda xx, tm: U.unet_forward(sd, cfg, xx, tm, z=zz)[0]
        for tv in (99, 0):
            tt = torch.full((N,), tv, dtype=torch.int64)
            eps = model_fn(x_t, sch.model_t(tt))
            o = D.ddim_step(sch, eps, x_t, tt)
            close(o["sample"].numpy(), g[f"ddim_step{tv}/sample"], 3e-5)
            # pred_xstart = sqrt(1/abar) x - sqrt(1/abar-1) eps amplifies eps rounding by up to ~150 at t'=99
            amp = float(sch.tab["sqrt_recipm1_alphas_cumprod"][tv])
            close(o["pred_xstart"].numpy(), g[f"ddim_step{tv}/pred_xstart"], 3e-5 + 3e-6 * amp)
            o = D.ddim_step(sch, eps, x_t, tt, torch.from_numpy(g[f"ddim_eta_step{tv}/noise"]), eta=0.7)
            close(o["sample"].numpy(), g[f"ddim_eta_step{tv}/sample"], 3e-5)
            o = D.p_sample_step(sch, eps, x_t, tt, torch.from_numpy(g[f"p_step{tv}/noise"]))
            close(o["sample"].numpy(), g[f"p_step{tv}/sample"], 3e-5)
            close(o["pred_xstart"].numpy(), g[f"p_step{tv}/pred_xstart"], 3e-5 + 3e-6 * amp)
        trace = []
        D.sample_loop(sch, model_fn, x_t, ddim=True, trace=trace)
        for k in (1, 2, 10, 50, 100):
            close(trace[k - 1].numpy(), g[f"loop/sample_after{k}"], 1e-4)


# ------------------------------------------------------------------ G9: learned sigma / variational bound
G9_VARIANTS = {"range": ("eps", "learned_range", True), "fixed": ("eps", "fixed_large", False),
               "xstart": ("xstart", "learned_range", True), "small": ("eps", "fixed_small", False)}


def g9_inputs():
    N = 4
    x0 = torch.round(synth("G9.x0", (N, 1, 28, 28), 0.0, 255.0)) / 127.5 - 1.0
    noise = synth("G9.noise", (N, 1, 28, 28), -1.7, 1.7)
    c = synth("G9.c", (N, 2), 0.0, 1.0)
    y = torch.tensor([1, 3, 5, 7], dtype=torch.int64)
    z = synth("G9.z", (N, 512), -1.0, 1.0)
    return N, x0, noise, c, y, z


def test_losses_py(golden):
    g = golden("g9_vlb.npz")
    sh = (3, 2, 8, 8)
    m1, lv1 = synth("G9.m1", sh, -1.0, 1.0), synth("G9.lv1", sh, -6.0, 0.5)
    m2, lv2 = synth("G9.m2", sh, -1.0, 1.0), synth("G9.lv2", sh, -6.0, 0.5)
    close(D.normal_kl(m1, lv1, m2, lv2).numpy(), g["losses/normal_kl"], 0.0, 1e-6)
    close(D.normal_kl(m1, lv1, 0.0, 0.0).numpy(), g["losses/normal_kl_scalar"], 0.0, 1e-6)
    xq = torch.from_numpy(g["losses/xq"])
    close(D.discretized_gaussian_log_likelihood(xq, m2, synth("G9.ls", sh, -4.0, 0.0)).numpy(), g["losses/dgll"], 1e-6, 1e-6)
    close(D.approx_standard_normal_cdf(synth("G9.cdf", (64,), -5.0, 5.0)).numpy(), g["losses/cdf"], 1e-7)


@pytest.mark.parametrize("tag", list(G9_VARIANTS))
def test_learned_sigma_and_bound(golden, tag):
    g = golden("g9_vlb.npz")
    mean_type, var_type, learn = G9_VARIANTS[tag]
    N, x0, noise, c, y, z = g9_inputs()
    t = torch.from_numpy(g["t"])
    sch = D.Schedule(1000, "linear", "", True)
    x_t = D.q_sample(sch, x0, t, noise)
    # the oracle network reproduces the reference's raw output (2 channels when sigma is learned)
    cfg = model_cfg("T28", learn_sigma=learn)
    sd = fill_state_dict(U.param_spec(cfg))
    with torch.no_grad():
        raw = U.unet_forward(sd, cfg, x_t, sch.model_t(t), y=y, c=c, z=z)[0]
    close(raw.numpy(), g[f"{tag}/vb_grad/raw"], 2e-5)
    raw = torch.from_numpy(g[f"{tag}/vb_grad/raw"])
    for clip in (True, False):
        pm = D.p_mean_variance_general(sch, raw, x_t, t, mean_type, var_type, clip)
        for k in ("mean", "variance", "log_variance", "pred_xstart"):
            close(pm[k].expand_as(x_t).numpy(), g[f"{tag}/pmv_clip{int(clip)}/{k}"], 1e-5, 1e-6)
        vb = D.vb_terms_bpd(sch, raw, x0, x_t, t, mean_type, var_type, clip)
        close(vb["output"].numpy(), g[f"{tag}/vb_clip{int(clip)}/output"], 1e-5, 1e-5)
        close(vb["pred_xstart"].numpy(), g[f"{tag}/vb_clip{int(clip)}/pred_xstart"], 1e-5, 1e-6)
    pm = D.p_mean_variance_general(sch, raw, x_t, t, mean_type, var_type, True)
    nz = (t != 0).float().reshape(-1, 1, 1, 1)
    close((pm["mean"] + nz * torch.exp(0.5 * pm["log_variance"]) * torch.from_numpy(g[f"{tag}/p_sample/noise"])).numpy(),
          g[f"{tag}/p_sample/sample"], 1e-5)
    close(D.prior_bpd(sch, x0).numpy(), g[f"{tag}/prior_bpd"], 1e-7, 1e-5)
    # gradient of the bound with respect to the raw output
    r = raw.clone().requires_grad_(True)
    out = D.vb_terms_bpd(sch, r, x0, x_t, t, mean_type, var_type, False)["output"]
    (out * torch.arange(1, N + 1, dtype=torch.float32)).sum().backward()
    close(r.grad.numpy(), g[f"{tag}/vb_grad/draw"], 1e-7, 1e-4)


def test_hybrid_loss_terms(golden):
    g = golden("g9_vlb.npz")
    N, x0, noise, c, y, z = g9_inputs()
    t = torch.from_numpy(g["t"])
    sch = D.Schedule(1000, "linear", "", True)
    cfg = model_cfg("T28", learn_sigma=True)
    spec = U.param_spec(cfg)
    sd = fill_state_dict(spec)
    pkeys = [k for k, _ in spec if "running" not in k and "num_batches" not in k]
    for k in pkeys:
        sd[k].requires_grad_(True)
    x_t = D.q_sample(sch, x0, t, noise)
    out, mu, var, zp, mask = U.unet_forward(sd, cfg, x_t, sch.model_t(t), y=y, c=c, x_start=x0,
                                            eps_z=torch.from_numpy(g["hybrid/eps_draw"]), training=True, new_stats={})
    terms = D.hybrid_losses(sch, out, x0, x_t, t, noise)
    terms["loss"].mean().backward()
    for k in ("mse", "vb", "loss"):
        close(terms[k].detach().numpy(), g[f"hybrid/{k}"], 1e-5, 1e-5)
    close(D.representation_loss(mu, var, zp, True, mask, c).detach().numpy(), g["hybrid/kld_rep"], 1e-4, 1e-5)
    sq = sum((sd[k].grad.double() ** 2).sum().item() for k in pkeys if sd[k].grad is not None)
    assert abs(sq - float(g["hybrid/grad_sqsum"])) <= 1e-3 * sq
    for k in ("out.2.weight", "out.2.bias", "input_blocks.1.0.in_layers.2.weight", "time_embed.0.weight"):
        probe_close(sd[k].grad, g, f"hybrid/grad/{k}", 1e-6, 1e-3)


def test_calc_bpd_loop(golden):
    g = golden("g9_vlb.npz")
    N, x0, noise, c, y, z = g9_inputs()
    sch = D.Schedule(1000, "linear", "8", True)
    cfg = model_cfg("T28", learn_sigma=True)
    sd = fill_state_dict(U.param_spec(cfg))

    def model_fn(x, tm):
        return U.unet_forward(sd, cfg, x, tm, y=y, c=c, z=z)[0]

    out = D.calc_bpd_loop(sch, model_fn, x0, torch.from_numpy(g["bpd/noise"]), "eps", "learned_range", True)
    for k in ("total_bpd", "prior_bpd", "vb", "xstart_mse", "mse"):
        close(out[k].numpy(), g[f"bpd/{k}"], 1e-5, 1e-4)


# ------------------------------------------------------------------ G10: flow_based=True (MultivariateCausalFlow)
def test_causal_flow(golden):
    g = golden("g10_flow.npz")
    cfg = model_cfg("T28", flow_based=True)
    spec = U.param_spec(cfg)
    assert [k for k, _ in spec] == list(g["keys"]) and [str(tuple(s)) for _, s in spec] == list(g["shapes"])
    sd = fill_state_dict(spec)
    mu = synth("G10.mu", (4, 512), -1.0, 1.0)
    C = torch.eye(2) - torch.tensor(U.ADJ["morpho"], dtype=torch.float32)
    with torch.no_grad():
        z, ld = U.causal_flow(sd, mu, C)
        rld, lp = U.causal_flow_reverse(sd, z, C)
    close(z.numpy(), g["flow/z_post"], 1e-5)
    close(ld.numpy(), g["flow/log_det"], 1e-4)
    close(rld.numpy(), g["flow/rev_log_det"], 1e-4)
    close(lp.numpy(), g["flow/log_prob"], 1e-3, 1e-6)


def test_flow_training_losses(golden):
    g = golden("g10_flow.npz")
    cfg = model_cfg("T28", flow_based=True)
    spec = U.param_spec(cfg)
    sd = fill_state_dict(spec)
    pkeys = [k for k, _ in spec if "running" not in k and "num_batches" not in k]
    for k in pkeys:
        sd[k].requires_grad_(True)
    N = 4
    x0 = synth("G10.x0", (N, 1, 28, 28), 0.0, 1.0)
    c = synth("G10.c", (N, 2), 0.0, 1.0)
    y = torch.tensor([0, 2, 4, 6], dtype=torch.int64)
    t = torch.tensor([3, 250, 600, 998], dtype=torch.int64)
    noise = synth("G10.noise", (N, 1, 28, 28), -1.7, 1.7)
    sch = D.Schedule(1000, "linear", "", True)
    eps_z = torch.from_numpy(g["eps_draw"])

    def model_full(x_t, tm, xs):
        return U.unet_forward(sd, cfg, x_t, tm, y=y, c=c, x_start=xs, eps_z=eps_z, training=True, new_stats={})

    terms = D.training_losses(sch, model_full, x0, t, noise, c=c, rep_cond=True, causal_modeling=True, kl_weight=0.5)
    terms["loss"].mean().backward()
    for k in ("loss", "mse", "kld_rep"):
        close(terms[k].detach().numpy(), g[f"train/{k}"], 1e-4, 1e-5)
    sq = sum((sd[k].grad.double() ** 2).sum().item() for k in pkeys if sd[k].grad is not None)
    assert abs(sq - float(g["train/grad_sqsum"])) <= 1e-3 * sq
    for k in ("causal_flow.s_cond.0.weight", "causal_flow.s_cond.4.bias", "causal_flow.t_cond.2.weight", "rep_emb.fc_mu.weight", "out.2.weight"):
        probe_close(sd[k].grad, g, f"train/grad/{k}", 1e-5, 2e-3)


# ------------------------------------------------------------------ G11: label-conditional and DiffAE model families
def g11_inputs():
    N = 3
    return (N, synth("G11.x0", (N, 1, 28, 28), 0.0, 1.0), synth("G11.c", (N, 2), 0.0, 1.0), synth("G11.c4", (N, 4), 0.0, 1.0),
            torch.tensor([2, 4, 9], dtype=torch.int64), synth("G11.noise", (N, 1, 28, 28), -1.7, 1.7),
            torch.tensor([10, 400, 990], dtype=torch.int64))


def test_label_conditional_family(golden):
    g = golden("g11_variants.npz")
    N, x0, c, c4, y, noise, tt = g11_inputs()
    cfg = U.default_cfg(**MODEL_CFG["T28"], context_cond=True)
    spec = U.param_spec(cfg)
    assert [k for k, _ in spec] == list(g["cond/keys"])
    sd = fill_state_dict(spec)
    for k, _ in spec:
        sd[k].requires_grad_(True)
    sch = D.Schedule(1000, "linear", "", True)
    terms = D.training_losses(sch, lambda x_t, tm, xs: U.unet_forward(sd, cfg, x_t, tm, y=y, c=c4), x0, tt, noise)
    terms["loss"].mean().backward()
    close(terms["loss"].detach().numpy(), g["cond/train/loss"], 1e-5, 1e-5)
    for k in ("c_emb.0.weight", "c_emb.2.bias", "out.2.weight"):
        probe_close(sd[k].grad, g, f"cond/train/grad/{k}", 1e-6, 1e-3)
    sch5 = D.Schedule(1000, "linear", "ddim5", True)
    cc = c4.clone()
    cc[:, 0] = -0.2
    x_t = D.q_sample(sch5, x0, torch.full((N,), sch5.T - 1, dtype=torch.int64), noise)
    with torch.no_grad():
        out = D.sample_loop(sch5, lambda x, tm: U.unet_forward(sd, cfg, x, tm, y=y, c=cc)[0], x_t, ddim=True)
    close(out.numpy(), g["cond/sample"], 5e-5)


def test_diffae_family(golden):
    g = golden("g11_variants.npz")
    N, x0, c, c4, y, noise, tt = g11_inputs()
    cfg = U.default_cfg(**MODEL_CFG["T28"], rep_cond=True)
    spec = U.param_spec(cfg)
    assert [k for k, _ in spec] == list(g["diffae/keys"])
    sd = fill_state_dict(spec)
    pkeys = [k for k, _ in spec if "running" not in k and "num_batches" not in k]
    for k in pkeys:
        sd[k].requires_grad_(True)
    sch = D.Schedule(1000, "linear", "", True)
    eps_z = torch.from_numpy(g["diffae/eps_draw"])
    terms = D.training_losses(sch, lambda x_t, tm, xs: U.unet_forward(sd, cfg, x_t, tm, y=y, x_start=xs, eps_z=eps_z, training=True, new_stats={}),
                              x0, tt, noise, c=c, rep_cond=True, causal_modeling=False, kl_weight=0.3)
    terms["loss"].mean().backward()
    for k in ("loss", "mse", "kld_rep"):
        close(terms[k].detach().numpy(), g[f"diffae/train/{k}"], 1e-4, 1e-5)
    for k in ("rep_emb.fc_var.weight", "up_emb.weight", "out.2.weight"):
        probe_close(sd[k].grad, g, f"diffae/train/grad/{k}", 1e-6, 1e-3)
    with torch.no_grad():
        mu, var = U.encode(sd, x0, U.n_encoder_layers(sd), False, None)
        mu[:, 256:512] = 0.4
        z = U.reparameterize(mu, torch.full_like(mu, 0.001), torch.from_numpy(g["diffae/z_eps"]))
        close(z.numpy(), g["diffae/z"], 1e-5)
        sch5 = D.Schedule(1000, "linear", "ddim5", True)
        x_t = D.q_sample(sch5, x0, torch.full((N,), sch5.T - 1, dtype=torch.int64), noise)
        out = D.sample_loop(sch5, lambda x, tm: U.unet_forward(sd, cfg, x, tm, y=y, z=z)[0], x_t, ddim=True)
    close(out.numpy(), g["diffae/sample"], 5e-5)


# ------------------------------------------------------------------ G12: one training step of the full 41 M / 93 M parameter models
def _model_inputs(tag, cfg, N):
    C, S_, nv = cfg["in_channels"], cfg["image_size"], cfg["n_vars"]
    x0 = synth(tag + ".x0", (N, C, S_, S_), 0.0, 1.0)
    c = synth(tag + ".c", (N, nv), 0.0, 1.0)
    y = torch.tensor([(3 * i + 1) % 10 for i in range(N)], dtype=torch.int64) if cfg["class_cond"] else None
    return x0, c, y


def grad_probe_close(t, g, prefix, rel):
    """head / strided samples of a gradient against the reference's, relative to that tensor's own largest entry"""
    f = t.detach().double().flatten()
    scale = float(g[prefix + "/absmax"])
    for key, got in (("head", f[:16]), ("strided", f[::4999])):
        d = np.abs(got.numpy() - g[prefix + "/" + key].astype(np.float64)).max()
        assert d <= rel * scale + 1e-12, f"{prefix}/{key}: {d:.3e} vs scale {scale:.3e}"
    ss = float(g[prefix + "/sumsq"])
    assert abs((f * f).sum().item() - ss) <= 1e-3 * ss + 1e-20, prefix


@pytest.mark.parametrize("tag", ["M32", "C64"])
def test_full_model_training_step(golden, tag):
    g = golden("g12_full_train.npz")
    cfg = model_cfg(tag)
    sd = fill_state_dict(U.param_spec(cfg))
    names = [str(k) for k in g[f"{tag}/grad_names"]]
    for k in names:
        sd[k].requires_grad_(True)
    x0, c, y = _model_inputs(tag + ".train", cfg, 2)
    t = torch.tensor([37, 990], dtype=torch.int64)
    noise = synth(tag + ".train.noise", tuple(x0.shape), -1.7, 1.7)
    eps_z = torch.from_numpy(g[f"{tag}/eps_draw"])
    new = {}
    fn = lambda x_t, tm, xs: U.unet_forward(sd, cfg, x_t, tm, y=y, c=c, x_start=xs, eps_z=eps_z, training=True, new_stats=new)
    terms = D.training_losses(D.Schedule(1000, "linear", "", True), fn, x0, t, noise, c=c, rep_cond=True, causal_modeling=True, kl_weight=0.3)
    terms["loss"].mean().backward()
    for k in ("loss", "mse", "kld_rep"):
        close(terms[k].detach().numpy(), g[f"{tag}/{k}"], 1e-4, 1e-5)
    sq = sum((sd[k].grad.double() ** 2).sum().item() for k in names)
    assert abs(sq - float(g[f"{tag}/grad_sqsum"])) <= 1e-4 * sq
    for k in names:
        if float(g[f"{tag}/g/{k}/absmax"]) == 0.0:
            assert float(sd[k].grad.abs().max()) == 0.0, k          # zero-initialised partners: exactly zero in the reference too
            continue
        if k.startswith("rep_emb.encoder.") and k.endswith(".0.bias"):
            continue            # a conv bias ahead of a batch-statistics BatchNorm: mathematically zero gradient, rounding noise on both sides
        grad_probe_close(sd[k].grad, g, f"{tag}/g/{k}", 2e-4)
    for k, v in new.items():
        close(v.numpy(), g[f"{tag}/after/{k}"], 1e-6, 1e-5)


# ------------------------------------------------------------------ G13: guidance w (conditional and z = 0 forwards blended)
def test_guided_ddim_step(golden):
    g = golden("g13_guidance.npz")
    cfg = model_cfg("P64")
    sd = fill_state_dict(U.param_spec(cfg))
    sch = D.Schedule(1000, "linear", "ddim100", True)
    N = 2
    x0 = synth("P64.x0", (N, 4, 64, 64), 0.0, 1.0)
    z = synth("P64.z", (N, 512))
    x_t = D.q_sample(sch, x0, torch.full((N,), 99, dtype=torch.int64), synth("P64.qnoise", (N, 4, 64, 64), -1.7, 1.7))
    fn = lambda x, tm, zz: U.unet_forward(sd, cfg, x, tm, z=zz)[0]
    with torch.no_grad():
        for tv in (99, 40):
            tt = torch.full((N,), tv, dtype=torch.int64)
            for w in (0.5, 2.0):
                eps = D.guided_eps(fn, x_t, sch.model_t(tt), z, w)
                o = D.ddim_step(sch, eps, x_t, tt)
                close(o["sample"].numpy(), g[f"t{tv}/w{w}/sample"], 2e-5)
                close(o["pred_xstart"].numpy(), g[f"t{tv}/w{w}/pred_xstart"], 2e-5)
                close(D.p_mean_variance(sch, eps, x_t, tt)["mean"].numpy(), g[f"t{tv}/w{w}/mean"], 2e-5)


# ------------------------------------------------------------------ G14: the ancestral loop end to end (p_sample_loop, 20 respaced steps)
def test_p_sample_loop_m32(golden):
    g = golden("g14_p_sample_loop.npz")
    cfg = model_cfg("M32")
    sd = fill_state_dict(U.param_spec(cfg))
    sch = D.Schedule(1000, "linear", "20", True)
    N = 2
    z = synth("M32.z", (N, 512))
    y = torch.tensor([(3 * i + 1) % 10 for i in range(N)], dtype=torch.int64)
    x_T = synth("M32.xT", (N, 1, 32, 32), -1.7, 1.7)
    noises = [torch.from_numpy(n) for n in g["step_noise"]]
    trace = []
    with torch.no_grad():
        final = D.sample_loop(sch, lambda x, tm: U.unet_forward(sd, cfg, x, tm, y=y, z=z)[0], x_T, ddim=False, noises=noises, trace=trace)
    for k in (1, 10, 20):
        close(trace[k - 1].numpy(), g[f"sample_after{k}"], 5e-5)
    close(final.numpy(), g["sample_after20"], 5e-5)


# ------------------------------------------------------------------ G17..G20: the long loops, trained-like weights, traversal
def _cf_start(sd, cfg, g, sch, N, t_last):
    x, x0, c, z, _ = model_inputs("P64", cfg, N)
    A = torch.tensor(U.ADJ["pendulum"], dtype=torch.float32)
    mu, _ = U.encode(sd, x0, U.n_encoder_layers(sd))
    z_post = U.nonlinearity_add_back_noise(sd, mu, U.causal_masking(mu, A, 4), 4)
    z_post[:, :128] = 0.2
    zz = U.reparameterize(z_post, torch.full_like(mu, 0.001), torch.from_numpy(g["eps_draw"]))
    close(zz.numpy(), g["z"], 1e-5)
    x_t = D.q_sample(sch, x0, torch.full((N,), t_last, dtype=torch.int64), synth("P64.qnoise", (N, 4, 64, 64), -1.7, 1.7))
    close(x_t.numpy(), g["x_t"], 1e-6)
    return x_t, zz


def test_ddim250_p64(golden):
    """BASELINE config [4]: the 250-step deterministic loop of the reference, restated (gaussian_diffusion.py:598-680)"""
    g = golden("g17_ddim250.npz")
    cfg = model_cfg("P64")
    sd = fill_state_dict(U.param_spec(cfg))
    sch = D.Schedule(1000, "linear", "ddim250", True)
    assert sch.T == 250
    with torch.no_grad():
        x_t, zz = _cf_start(sd, cfg, g, sch, 2, 249)
        trace = []
        D.sample_loop(sch, lambda xx, tm: U.unet_forward(sd, cfg, xx, tm, z=zz)[0], x_t, ddim=True, trace=trace)
    for k in (1, 25, 125, 250):
        close(trace[k - 1].numpy(), g[f"sample_after{k}"], 1e-4)


def test_p_sample_loop_t1000_m32(golden):
    """BASELINE config [0]: 1000 ancestral steps (gaussian_diffusion.py:416-504), per-step noise in closed form"""
    g = golden("g18_p_sample_t1000.npz")
    cfg = model_cfg("M32")
    sd = fill_state_dict(U.param_spec(cfg))
    sch = D.Schedule(1000, "linear", "", True)
    N = 2
    z = synth("M32.z", (N, 512))
    y = torch.tensor([(3 * i + 1) % 10 for i in range(N)], dtype=torch.int64)
    x_T = synth("M32.xT", (N, 1, 32, 32), -1.7, 1.7)

    class Noise:
        def __getitem__(self, k):
            return synth_noise(f"G18.noise.{k}", (N, 1, 32, 32))

    trace = []
    with torch.no_grad():
        D.sample_loop(sch, lambda x, tm: U.unet_forward(sd, cfg, x, tm, y=y, z=z)[0], x_T, ddim=False, noises=Noise(), trace=trace)
    for k in (1, 10, 100, 500, 900, 1000):
        close(trace[k - 1].numpy(), g[f"sample_after{k}"], 1e-4)


def test_trained_like_weights_p64(golden):
    """P64 with log-uniform weight magnitudes (closed_form.fill_value_trained): forward, one step and the DDIM-100 loop"""
    g = golden("g19_trained_like.npz")
    cfg = model_cfg("P64")
    sd = fill_state_dict_trained(U.param_spec(cfg))
    assert [k for k, v in sd.items() if v.dim() >= 2 and float(v.abs().max()) == 0.0] == list(g["zero_keys"])
    sch = D.Schedule(1000, "linear", "ddim100", True)
    N = 2
    x, x0, c, z, _ = model_inputs("P64", cfg, N)
    with torch.no_grad():
        e = U.unet_forward(sd, cfg, x, torch.tensor([37.0, 990.0]), z=z)[0]
        close(e.numpy(), g["eps_z"], 3e-5)
        x_t, zz = _cf_start(sd, cfg, g, sch, N, 99)
        model_fn = lambda xx, tm: U.unet_forward(sd, cfg, xx, tm, z=zz)[0]
        t99 = torch.full((N,), 99, dtype=torch.int64)
        o = D.ddim_step(sch, model_fn(x_t, sch.model_t(t99)), x_t, t99)
        close(o["sample"].numpy(), g["step99/sample"], 3e-5)
        trace = []
        D.sample_loop(sch, model_fn, x_t, ddim=True, trace=trace)
    for k in (1, 10, 50, 100):
        close(trace[k - 1].numpy(), g[f"loop/sample_after{k}"], 1e-4)


def test_traversal_p64(golden):
    """The script's traversal (image_causaldae_test.py:481-531) restated on the oracle: the eight conditioning vectors z (mu[:, 16:32]
    := the accumulated value, causal layer, fresh draw) for every value, and the whole DDIM-250 decode for the LAST value (one of
    eight: the CPU suite stays within minutes; the GPU test decodes all eight)."""
    g = golden("g20_traversal.npz")
    cfg = model_cfg("P64")
    sd = fill_state_dict(U.param_spec(cfg))
    sch = D.Schedule(1000, "linear", "ddim250", True)
    N = 2
    x, x0, c, z, _ = model_inputs("P64", cfg, N)
    A = torch.tensor(U.ADJ["pendulum"], dtype=torch.float32)
    value, zs = -0.5, []
    with torch.no_grad():
        x_t = D.q_sample(sch, x0, torch.full((N,), 249, dtype=torch.int64), synth("P64.qnoise", (N, 4, 64, 64), -1.7, 1.7))
        close(x_t.numpy(), g["x_t"], 1e-6)
        for i in range(8):
            assert value == float(g["values"][i])
            mu, _ = U.encode(sd, x0, U.n_encoder_layers(sd))
            mu[:, 16:32] = value
            z_post = U.nonlinearity_add_back_noise(sd, mu, U.causal_masking(mu, A, 4), 4)
            zz = U.reparameterize(z_post, torch.full_like(mu, 0.001), torch.from_numpy(g["eps_draws"][i]))
            close(zz.numpy(), g[f"z{i}"], 1e-5)
            zs.append(zz)
            value += 0.15
        final = D.sample_loop(sch, lambda xx, tm: U.unet_forward(sd, cfg, xx, tm, z=zs[7])[0], x_t, ddim=True)
    close(final.numpy(), g["sample7"], 1e-4)


# ------------------------------------------------------------------ G21: the converging run the 16-bit torso is trained against
def test_training_curve_m32(golden):
    """The first 8 of G21's 24 reference steps (M32, batch 16, AdamW 1e-4, kl 0.1, four batches in rotation; reference train_util.py:231-297)
    through the oracle's training_losses + adamw_ema_step: the loss curve the GPU test holds the parity mode and the 16-bit torso to."""
    g = golden("g21_m32_curve.npz")
    cfg = model_cfg("M32")
    spec = U.param_spec(cfg)
    sd = fill_state_dict(spec)
    pkeys = [k for k, _ in spec if "running" not in k and "num_batches" not in k]
    params = [sd[k].requires_grad_(True) for k in pkeys]
    ema = [p.detach().clone() for p in params]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    sch = D.Schedule(1000, "linear", "", True)
    N = int(g["batch"])
    for step in range(8):
        b = step % 4
        x0 = synth(f"M32c.{b}.x0", (N, 1, 32, 32), -1.0, 1.0)
        c = synth(f"M32c.{b}.c", (N, 2), 0.0, 1.0)
        y = torch.tensor([(b + 3 * i) % 10 for i in range(N)], dtype=torch.int64)
        t = torch.tensor([(137 * (step + 1) + 251 * i) % 1000 for i in range(N)], dtype=torch.int64)
        noise = synth_noise(f"M32c.{step}.noise", (N, 1, 32, 32))
        torch.manual_seed(300 + step)
        eps_z = torch.randn(N, 512)
        chk = g[f"step{step}/eps_draw_check"]
        assert abs(eps_z.double().sum().item() - chk[0]) < 1e-6 * max(1.0, abs(chk[0])) and np.allclose(eps_z.flatten()[:6].numpy(), chk[2:], atol=0)
        new = {}

        def model_full(x_t, tm, xs):
            return U.unet_forward(sd, cfg, x_t, tm, y=y, c=c, x_start=xs, eps_z=eps_z, training=True, new_stats=new)

        for p in params:
            p.grad = None
        terms = D.training_losses(sch, model_full, x0, t, noise, c=c, rep_cond=True, causal_modeling=True, kl_weight=0.1)
        terms["loss"].mean().backward()
        for k in ("loss", "mse", "kld_rep"):
            ref = float(g[f"step{step}/{k}_mean"])
            got = float(terms[k].detach().double().mean())
            assert abs(got - ref) <= 2e-5 * abs(ref), (step, k, got, ref)
        with torch.no_grad():
            D.adamw_ema_step(params, [p.grad for p in params], m, v, ema, step + 1)
            for k, val in new.items():
                sd[k] = val
    assert float(g["step0/loss_mean"]) - float(g["step23/loss_mean"]) > 4.0          # (the fixture's run converges)


# ------------------------------------------------------------------ G22: resampling without a conv
def test_plain_resample(golden):
    """Upsample(use_conv=False) = nearest 2x alone (reference unet.py:76-78): the fixture's output and gradient are what nearest
    interpolation and its 2 x 2 sum give; and the fixture records that the reference's Downsample(use_conv=False) cannot be built."""
    g = golden("g22_plain_resample.npz")
    x = synth("T28r.x", (2, 64, 7, 14))
    gy = synth("T28r.gy", (2, 64, 14, 28))
    close(x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3).numpy(), g["up/y"], 0.0)
    close(gy.reshape(2, 64, 7, 2, 14, 2).sum(dim=(3, 5)).numpy(), g["up/dx"], 1e-6)
    assert int(g["down/reference_builds"]) == 0 and "kernel_size" in str(g["down/error"])

