"""Counterfactual generation: encode -> causal layer -> do-intervention on one variable's latent slice -> q_sample to the
last (spaced) step -> DDIM / ancestral decode.  This is the call pattern of the reference's evaluation script
(scripts/image_causaldae_test.py:405-436 MorphoMNIST, :535-594 pendulum, :773-815 circuit; traversal :481-531), factored
into functions; the image batch is sharded over ranks with no collective inside the loop and gathered at the end.
The two sibling scripts use the same kernels with other conditioning: image_diffae_test.py (representation without the
causal layer: the edit goes into mu) is handled by `counterfactual_sample` on a `causal_modeling=False` model, and
image_conditional_test.py (label vector `c` through c_emb) by `label_conditional_sample`."""
import torch as th

from . import dist_util
from .nn import reparameterize
from .unet import ADJACENCY


def encode_with_intervention(model, batch, A, var_index=None, value=None, var=0.001, eps=None, intervene_on="z_post", columns=None):
    """-> z [N, rep_dim] conditioning the decoder.

    mu = encoder mean of `batch`; z_pre = A^T mu; z_post = MLP(z_pre) + mu; then the slice of variable `var_index`
    (rep_dim / n_vars wide; or the explicit column range `columns` = (lo, hi), as the script's traversal hard-codes one) is
    overwritten with `value` — on z_post (pendulum / circuit branches of the script) or on mu before the causal layer
    (`intervene_on="mu"`, the MorphoMNIST branch and the traversal) — and z = z_post + sqrt(var) * eps."""
    mu, _ = model.rep_emb.encode(batch)
    nv = model.n_vars
    d = mu.shape[1] // nv
    sl = None
    if columns is not None:
        sl = slice(int(columns[0]), int(columns[1]))
    elif var_index is not None:
        sl = slice(var_index * d, (var_index + 1) * d)
    if not getattr(model, "causal_modeling", True):      # DiffAE (image_diffae_test.py:270-300): no causal layer, edit mu itself
        if sl is not None:
            mu = mu.clone()
            mu[:, sl] = value
        return reparameterize(mu, th.full_like(mu, var), eps=eps)
    A = th.as_tensor(A, dtype=th.float32)
    if sl is not None and intervene_on == "mu":
        mu = mu.clone()
        mu[:, sl] = value
    z_pre = model.causal_mask.causal_masking(mu, A)
    z_post = model.causal_mask.nonlinearity_add_back_noise(mu, z_pre)
    if sl is not None and intervene_on == "z_post":
        z_post[:, sl] = value
    return reparameterize(z_post, th.full_like(z_post, var), eps=eps)


def counterfactual_sample(model, diffusion, batch, A="circuit", var_index=None, value=None, *, use_ddim=True, eta=0.0, w=None,
                          clip_denoised=True, extra_kwargs=None, q_noise=None, z_eps=None, intervene_on="z_post",
                          use_graph=True, shard=False, gather=False, columns=None):
    """Counterfactual images for `batch` [N,C,S,S] (values in the training range) under do(var_index := value).

    Returns the decoded samples (this rank's shard unless gather=True).  `shard=True` splits the batch over the ranks of
    the default process group (`dist_util.shard_range`); no collective runs inside the sampling loop."""
    if isinstance(A, str):
        A = ADJACENCY[A]
    if shard:
        lo, hi = dist_util.shard_range(batch.shape[0])
        batch = batch[lo:hi]
        q_noise = None if q_noise is None else q_noise[lo:hi]
        z_eps = None if z_eps is None else z_eps[lo:hi]
    dev = next(model.parameters()).device
    batch = batch.to(dev)
    with th.no_grad():
        z = encode_with_intervention(model, batch, A, var_index, value, eps=z_eps, intervene_on=intervene_on, columns=columns)
        t_last = th.full((batch.shape[0],), diffusion.num_timesteps - 1, dtype=th.int64, device=dev)
        noise = th.randn_like(batch) if q_noise is None else q_noise.to(dev)
        x_t = diffusion.q_sample(batch, t_last, noise=noise)          # the script starts from a noised input, not pure noise
        kw = dict(extra_kwargs or {})
        kw["z"] = z
        if use_ddim:
            sample = diffusion.ddim_sample_loop(model, tuple(batch.shape), noise=x_t, clip_denoised=clip_denoised, model_kwargs=kw,
                                                eta=eta, w=w, use_graph=use_graph and eta == 0.0)
        else:
            sample = diffusion.p_sample_loop(model, tuple(batch.shape), noise=x_t, clip_denoised=clip_denoised, model_kwargs=kw)
    if gather:
        return th.cat(dist_util.gather_samples(sample), dim=0)
    return sample


def traversal_values(start=-0.5, step=0.15, count=8):
    """The intervention values of the script's traversal: `value = -0.5`, then `value += 0.15` eight times in a Python float
    (image_causaldae_test.py:503-529) — the ACCUMULATED doubles, not start + i * step."""
    out, v = [], float(start)
    for _ in range(count):
        out.append(v)
        v += step
    return out


def latent_traversal(model, diffusion, batch, A, var_index=None, values=None, *, columns=None, intervene_on=None, q_noise=None, z_eps=None, **kw):
    """One counterfactual batch per intervention value: the reference's traversal loop (image_causaldae_test.py:481-531).

    As there: the noised start x_t = q_sample(batch, T - 1, noise) is computed ONCE and shared by every value (one noise draw, or
    `q_noise`); every value gets a FRESH reparameterisation draw (`z_eps`: None, one tensor for all, or a list with one per value).
    Called with neither `var_index` nor `columns` it is the script itself: `mu[:, 16:32] = value` before the causal layer for
    value = -0.5, -0.35, ... (traversal_values()).  With `var_index` the edit goes to that variable's rep_dim / n_vars wide slice, on
    z_post unless `intervene_on="mu"`."""
    if var_index is None and columns is None:
        columns = (16, 32)                       # the script's hard-coded slice
        intervene_on = intervene_on or "mu"
    intervene_on = intervene_on or "z_post"
    values = traversal_values() if values is None else [float(v) for v in values]
    if q_noise is None:
        q_noise = th.randn_like(batch)
    if not isinstance(z_eps, (list, tuple)):
        z_eps = [z_eps] * len(values)
    assert len(z_eps) == len(values)
    return [counterfactual_sample(model, diffusion, batch, A, var_index, v, columns=columns, intervene_on=intervene_on, q_noise=q_noise, z_eps=e, **kw)
            for v, e in zip(values, z_eps)]


def label_conditional_sample(model, diffusion, batch, cond, var_index=None, value=None, *, from_input=True, use_ddim=True, eta=0.0,
                             clip_denoised=True, q_noise=None, use_graph=True, shard=False, gather=False):
    """Label-conditional generation (scripts/image_conditional_test.py:112-150, 224-240): the model is conditioned on the
    label vector cond["c"] (and cond["y"] if class-conditional); do(var_index := value) overwrites one label column.
    from_input=True starts the reverse chain from q_sample(batch, T-1) like the script's pendulum branch; False from noise."""
    cond = {k: v.clone() for k, v in cond.items()}
    if shard:
        lo, hi = dist_util.shard_range(batch.shape[0])
        batch = batch[lo:hi]
        cond = {k: v[lo:hi] for k, v in cond.items()}
        q_noise = None if q_noise is None else q_noise[lo:hi]
    dev = next(model.parameters()).device
    batch = batch.to(dev)
    cond = {k: v.to(dev) for k, v in cond.items()}
    if var_index is not None:
        cond["c"][:, var_index] = value
    with th.no_grad():
        start = None
        if from_input:
            t_last = th.full((batch.shape[0],), diffusion.num_timesteps - 1, dtype=th.int64, device=dev)
            start = diffusion.q_sample(batch, t_last, noise=th.randn_like(batch) if q_noise is None else q_noise.to(dev))
        if use_ddim:
            sample = diffusion.ddim_sample_loop(model, tuple(batch.shape), noise=start, clip_denoised=clip_denoised, model_kwargs=cond,
                                                eta=eta, use_graph=use_graph and eta == 0.0)
        else:
            sample = diffusion.p_sample_loop(model, tuple(batch.shape), noise=start, clip_denoised=clip_denoised, model_kwargs=cond)
    if gather:
        return th.cat(dist_util.gather_samples(sample), dim=0)
    return sample
