"""Oracle: diffusion math — q_sample, DDIM / ancestral steps and loops, losses, AdamW/EMA.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Every random draw is an explicit
argument.  `model_fn(x, t_model)` returns eps only; `Schedule.model_t` maps the
spaced step index to what the network sees (respace.py:119-124).

Reference lines restated:
  _extract_into_tensor                 gaussian_diffusion.py:938-951
  q_sample                             gaussian_diffusion.py:201-222
  p_mean_variance (eps / FIXED_LARGE)  gaussian_diffusion.py:248-353
  p_sample                             gaussian_diffusion.py:383-414
  ddim_sample                          gaussian_diffusion.py:506-558
  *_sample_loop                        gaussian_diffusion.py:416-504, 598-680
  prior / representation_loss          gaussian_diffusion.py:718-766, nn.py:440-457
  training_losses (MSE branch)         gaussian_diffusion.py:768-859
  TrainLoop optimizer step             train_util.py:176-187,210-214,292-311; nn.py:503-513
  learned-sigma p_mean_variance        gaussian_diffusion.py:289-303
  _vb_terms_bpd / _prior_bpd / calc_bpd_loop   gaussian_diffusion.py:682-715, 862-931
  normal_kl / discretized Gaussian LL  losses.py:12-77
  hybrid + bound-only training losses  gaussian_diffusion.py:792-837 (frozen-mean lambda fixed to a 5-tuple, see DESIGN Q8)
"""
import numpy as np
import torch

from . import schedule as S


class Schedule:
    def __init__(self, steps=1000, noise_schedule="linear", timestep_respacing="", rescale_timesteps=True):
        self.tab, self.timestep_map, self.original_T = S.make_schedule(steps, noise_schedule, timestep_respacing)
        self.T = len(self.tab["betas"])
        self.rescale = rescale_timesteps

    def ext(self, name, t, ndim=4):
        """f64 table gathered at int64 t, cast to fp32, shaped for broadcast."""
        v = torch.from_numpy(self.tab[name])[t].float()
        return v.reshape(-1, *([1] * (ndim - 1)))

    def model_t(self, t):
        """respace.py:119-124: spaced index -> original index (int64, bit exact) -> float * 1000/T."""
        m = torch.tensor(self.timestep_map, dtype=t.dtype)[t]
        return m.float() * (1000.0 / self.original_T) if self.rescale else m


def q_sample(sch, x0, t, noise):
    return sch.ext("sqrt_alphas_cumprod", t, x0.dim()) * x0 + sch.ext("sqrt_one_minus_alphas_cumprod", t, x0.dim()) * noise


def p_mean_variance(sch, eps, x, t, clip=True):
    nd = x.dim()
    x0 = sch.ext("sqrt_recip_alphas_cumprod", t, nd) * x - sch.ext("sqrt_recipm1_alphas_cumprod", t, nd) * eps
    if clip:
        x0 = x0.clamp(-1, 1)
    mean = sch.ext("posterior_mean_coef1", t, nd) * x0 + sch.ext("posterior_mean_coef2", t, nd) * x
    return dict(mean=mean, pred_xstart=x0,
                variance=sch.ext("fixed_large_variance", t, nd).expand_as(x),
                log_variance=sch.ext("fixed_large_log_variance", t, nd).expand_as(x))


def guided_eps(model_fn, x, tm, z, w, rep_dim=512):
    """gaussian_diffusion.py:277-285 with the reference's zeros(N,64) bug fixed to rep_dim."""
    e_c = model_fn(x, tm, z)
    if w is None:
        return e_c
    e_u = model_fn(x, tm, torch.zeros(x.shape[0], rep_dim))
    return w * e_c + (1 - w) * e_u


def p_sample_step(sch, eps, x, t, noise, clip=True):
    out = p_mean_variance(sch, eps, x, t, clip)
    nz = (t != 0).float().reshape(-1, *([1] * (x.dim() - 1)))
    return dict(sample=out["mean"] + nz * torch.exp(0.5 * out["log_variance"]) * noise, pred_xstart=out["pred_xstart"])


def ddim_step(sch, eps, x, t, noise=None, eta=0.0, clip=True):
    nd = x.dim()
    x0 = p_mean_variance(sch, eps, x, t, clip)["pred_xstart"]
    eps2 = (sch.ext("sqrt_recip_alphas_cumprod", t, nd) * x - x0) / sch.ext("sqrt_recipm1_alphas_cumprod", t, nd)
    ab, abp = sch.ext("alphas_cumprod", t, nd), sch.ext("alphas_cumprod_prev", t, nd)
    sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
    mean = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps2
    nz = (t != 0).float().reshape(-1, *([1] * (nd - 1)))
    if noise is None:
        noise = torch.zeros_like(x)
    return dict(sample=mean + nz * sigma * noise, pred_xstart=x0)


def sample_loop(sch, model_fn, x_T, ddim=True, eta=0.0, clip=True, noises=None, n_steps=None, trace=None):
    """model_fn(x, t_model) -> eps.  noises[i] = draw used at loop iteration i (None => zeros)."""
    img = x_T
    idx = list(range(sch.T))[::-1]
    if n_steps is not None:
        idx = idx[:n_steps]
    for k, i in enumerate(idx):
        t = torch.full((x_T.shape[0],), i, dtype=torch.int64)
        with torch.no_grad():
            eps = model_fn(img, sch.model_t(t))
            nz = None if noises is None else noises[k]
            out = ddim_step(sch, eps, img, t, nz, eta, clip) if ddim else p_sample_step(sch, eps, img, t, nz, clip)
        img = out["sample"]
        if trace is not None:
            trace.append(img.clone())
    return img


def kl_normal(qm, qv, pm, pv):
    return (0.5 * (torch.log(pv) - torch.log(qv) + qv / pv + (qm - pm).pow(2) / pv - 1)).sum(-1)


def representation_loss(mu, var, z_post, causal_modeling, mask, c):
    """gaussian_diffusion.py:727-766; prior mean of variable i is c[:, i] broadcast (scale [[0,1]]*nv)."""
    nv = c.shape[1]
    kld = kl_normal(mu, var, torch.zeros_like(mu), torch.ones_like(var))
    if causal_modeling:
        d = mu.shape[1] // nv
        zp = z_post.reshape(-1, nv, d)
        one = torch.ones(mu.shape[0], d)
        for i in range(nv):
            pm = c[:, i].float()[:, None].expand(-1, d)
            kld = kld + kl_normal(zp[:, i], one, pm, one)
    if mask is not None:
        kld = (kld * mask).sum() / mask.sum()
    return kld


def training_losses(sch, model_full, x0, t, noise, c=None, rep_cond=False, causal_modeling=False, kl_weight=0.0):
    """model_full(x_t, t_model, x_start) -> (eps, mu, var, z_post, mask)."""
    x_t = q_sample(sch, x0, t, noise)
    eps, mu, var, z_post, mask = model_full(x_t, sch.model_t(t), x0)
    terms = {}
    if rep_cond:
        terms["kld_rep"] = representation_loss(mu, var, z_post, causal_modeling, mask, c)
    terms["mse"] = ((noise - eps) ** 2).mean(dim=list(range(1, x0.dim())))
    terms["loss"] = terms["mse"] + kl_weight * terms["kld_rep"] if rep_cond else terms["mse"]
    return terms


# --------------------------------------------------------------------------- learned sigma / variational bound
def normal_kl(mean1, logvar1, mean2, logvar2):
    """losses.py:12-39."""
    logvar1, logvar2 = [v if isinstance(v, torch.Tensor) else torch.tensor(float(v)) for v in (logvar1, logvar2)]
    return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * torch.exp(-logvar2))


def approx_standard_normal_cdf(x):
    """losses.py:42-47."""
    return 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * torch.pow(x, 3))))


def discretized_gaussian_log_likelihood(x, means, log_scales):
    """losses.py:50-77."""
    cen = x - means
    inv = torch.exp(-log_scales)
    cdf_plus = approx_standard_normal_cdf(inv * (cen + 1.0 / 255.0))
    cdf_min = approx_standard_normal_cdf(inv * (cen - 1.0 / 255.0))
    log_cdf_plus = torch.log(cdf_plus.clamp(min=1e-12))
    log_one_minus = torch.log((1.0 - cdf_min).clamp(min=1e-12))
    delta = cdf_plus - cdf_min
    return torch.where(x < -0.999, log_cdf_plus, torch.where(x > 0.999, log_one_minus, torch.log(delta.clamp(min=1e-12))))


def p_mean_variance_general(sch, model_out, x, t, mean_type="eps", var_type="fixed_large", clip=True):
    """gaussian_diffusion.py:289-353.  mean_type: "eps" | "xstart"; var_type: "fixed_large" | "fixed_small" | "learned" |
    "learned_range".  model_out is [N, C or 2C, ...]."""
    nd = x.dim()
    C = x.shape[1]
    if var_type in ("learned", "learned_range"):
        model_out, vals = torch.split(model_out, C, dim=1)
        if var_type == "learned":
            logvar = vals
        else:
            min_log = sch.ext("posterior_log_variance_clipped", t, nd)
            max_log = torch.from_numpy(np.log(sch.tab["betas"]))[t].float().reshape(-1, *([1] * (nd - 1)))
            frac = (vals + 1) / 2
            logvar = frac * max_log + (1 - frac) * min_log
        var = torch.exp(logvar)
    elif var_type == "fixed_large":
        var, logvar = sch.ext("fixed_large_variance", t, nd).expand_as(x), sch.ext("fixed_large_log_variance", t, nd).expand_as(x)
    else:
        var, logvar = sch.ext("posterior_variance", t, nd).expand_as(x), sch.ext("posterior_log_variance_clipped", t, nd).expand_as(x)
    if mean_type == "xstart":
        x0 = model_out
    else:
        x0 = sch.ext("sqrt_recip_alphas_cumprod", t, nd) * x - sch.ext("sqrt_recipm1_alphas_cumprod", t, nd) * model_out
    if clip:
        x0 = x0.clamp(-1, 1)
    mean = sch.ext("posterior_mean_coef1", t, nd) * x0 + sch.ext("posterior_mean_coef2", t, nd) * x
    return dict(mean=mean, variance=var, log_variance=logvar, pred_xstart=x0)


def vb_terms_bpd(sch, model_out, x0, x_t, t, mean_type="eps", var_type="fixed_large", clip=True):
    """gaussian_diffusion.py:682-715."""
    nd = x0.dim()
    true_mean = sch.ext("posterior_mean_coef1", t, nd) * x0 + sch.ext("posterior_mean_coef2", t, nd) * x_t
    true_lv = sch.ext("posterior_log_variance_clipped", t, nd)
    out = p_mean_variance_general(sch, model_out, x_t, t, mean_type, var_type, clip)
    dims = list(range(1, nd))
    kl = normal_kl(true_mean, true_lv, out["mean"], out["log_variance"]).mean(dim=dims) / np.log(2.0)
    nll = (-discretized_gaussian_log_likelihood(x0, out["mean"], 0.5 * out["log_variance"])).mean(dim=dims) / np.log(2.0)
    return dict(output=torch.where(t == 0, nll, kl), pred_xstart=out["pred_xstart"])


def prior_bpd(sch, x0):
    """gaussian_diffusion.py:862-880."""
    t = torch.full((x0.shape[0],), sch.T - 1, dtype=torch.int64)
    mean = sch.ext("sqrt_alphas_cumprod", t, x0.dim()) * x0
    lv = sch.ext("log_one_minus_alphas_cumprod", t, x0.dim())
    return normal_kl(mean, lv, 0.0, 0.0).mean(dim=list(range(1, x0.dim()))) / np.log(2.0)


def calc_bpd_loop(sch, model_fn, x0, noises, mean_type="eps", var_type="fixed_large", clip=True):
    """gaussian_diffusion.py:882-931; model_fn(x, t_model) -> raw model output; noises[k] is the draw of loop iteration k."""
    nd = x0.dim()
    dims = list(range(1, nd))
    vb, xs_mse, mse = [], [], []
    for k, i in enumerate(list(range(sch.T))[::-1]):
        t = torch.full((x0.shape[0],), i, dtype=torch.int64)
        x_t = q_sample(sch, x0, t, noises[k])
        with torch.no_grad():
            out = vb_terms_bpd(sch, model_fn(x_t, sch.model_t(t)), x0, x_t, t, mean_type, var_type, clip)
        vb.append(out["output"])
        xs_mse.append(((out["pred_xstart"] - x0) ** 2).mean(dim=dims))
        eps = (sch.ext("sqrt_recip_alphas_cumprod", t, nd) * x_t - out["pred_xstart"]) / sch.ext("sqrt_recipm1_alphas_cumprod", t, nd)
        mse.append(((eps - noises[k]) ** 2).mean(dim=dims))
    vb, xs_mse, mse = torch.stack(vb, 1), torch.stack(xs_mse, 1), torch.stack(mse, 1)
    pb = prior_bpd(sch, x0)
    return dict(total_bpd=vb.sum(1) + pb, prior_bpd=pb, vb=vb, xstart_mse=xs_mse, mse=mse)


def hybrid_losses(sch, model_out, x0, x_t, t, noise, var_type="learned_range", rescaled=True):
    """gaussian_diffusion.py:813-850 for learned variances: mse on the eps half + the bound on [eps.detach() | var half]
    (x T/1000 for RESCALED_MSE)."""
    C = x0.shape[1]
    eps_out, var_out = torch.split(model_out, C, dim=1)
    frozen = torch.cat([eps_out.detach(), var_out], dim=1)
    vb = vb_terms_bpd(sch, frozen, x0, x_t, t, "eps", var_type, clip=False)["output"]
    if rescaled:
        vb = vb * (sch.T / 1000.0)
    mse = ((noise - eps_out) ** 2).mean(dim=list(range(1, x0.dim())))
    return dict(mse=mse, vb=vb, loss=mse + vb)


def kl_weight_at(step, total=50000):
    """train_util.py:176-187 linear 0->1 over `total` steps (t = step/(total-1))."""
    if step >= total:
        return 1.0
    if step <= 0:
        return 0.0
    return step / (total - 1)


def adamw_ema_step(params, grads, m, v, ema, step, lr=1e-4, wd=0.0, b1=0.9, b2=0.999, eps=1e-8, rate=0.9999):
    """torch.optim.AdamW semantics (decoupled decay, bias correction) + update_ema (nn.py:503-513). In place."""
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for p, g, mi, vi, e in zip(params, grads, m, v, ema):
        p.mul_(1 - lr * wd)
        mi.mul_(b1).add_(g, alpha=1 - b1)
        vi.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (vi.sqrt() / np.sqrt(bc2)).add_(eps)
        p.addcdiv_(mi, denom, value=-lr / bc1)
        e.mul_(rate).add_(p, alpha=1 - rate)
