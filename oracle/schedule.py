"""Oracle: noise-schedule tables and timestep respacing (numpy float64 / python ints).

TEST INFRASTRUCTURE — see oracle/__init__.py.  Restates
  * get_named_beta_schedule        reference gaussian_diffusion.py:21-45
  * GaussianDiffusion.__init__     reference gaussian_diffusion.py:121-182
  * space_timesteps                reference respace.py:7-61
  * SpacedDiffusion.__init__       reference respace.py:74-88
  * FIXED_LARGE variance tables    reference gaussian_diffusion.py:305-311
"""
import math

import numpy as np

TABLE_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1",
    "posterior_mean_coef2",
)


def named_betas(name, T):
    """gaussian_diffusion.py:30-43 (linear: endpoints scaled by 1000/T; cosine: alpha-bar ratio capped at .999)."""
    if name == "linear":
        k = 1000.0 / T
        return np.linspace(k * 1e-4, k * 2e-2, T, dtype=np.float64)
    if name == "cosine":
        ab = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1.0 - ab((i + 1) / T) / ab(i / T), 0.999) for i in range(T)])
    raise NotImplementedError(name)


def tables(betas):
    """All f64 coefficient tables of gaussian_diffusion.py:137-179, keyed by the reference attribute name."""
    b = np.asarray(betas, dtype=np.float64)
    assert b.ndim == 1 and (b > 0).all() and (b <= 1).all()
    a = 1.0 - b
    ac = np.cumprod(a)
    ac_prev = np.concatenate([[1.0], ac[:-1]])
    ac_next = np.concatenate([ac[1:], [0.0]])
    pv = b * (1.0 - ac_prev) / (1.0 - ac)
    out = {
        "betas": b,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "alphas_cumprod_next": ac_next,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": np.log(np.concatenate([pv[1:2], pv[1:]])),
        "posterior_mean_coef1": b * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(a) / (1.0 - ac),
    }
    # FIXED_LARGE model variance (gaussian_diffusion.py:308-311)
    fl = np.concatenate([pv[1:2], b[1:]])
    out["fixed_large_variance"] = fl
    out["fixed_large_log_variance"] = np.log(fl)
    return out


def space_timesteps(T, spec):
    """respace.py:7-61.  Returns a python set of ints (bit-exact requirement).

    "ddimN": the first integer stride whose range has exactly N entries.
    "a,b,c" / list: T is cut into len(spec) sections; each section contributes
    `count` steps at a fractional stride accumulated in a python float and
    rounded with python's round() (banker's rounding).
    """
    if isinstance(spec, str):
        if spec.startswith("ddim"):
            want = int(spec[4:])
            for stride in range(1, T):
                if len(range(0, T, stride)) == want:
                    return set(range(0, T, stride))
            raise ValueError(f"cannot create exactly {T} steps with an integer stride")
        spec = [int(s) for s in spec.split(",")]
    nsec = len(spec)
    base, extra = divmod(T, nsec)
    steps, start = [], 0
    for i, count in enumerate(spec):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            steps.append(start + round(pos))
            pos += stride
        start += size
    return set(steps)


def respaced_betas(base_betas, use_timesteps):
    """respace.py:79-87: beta'_i = 1 - abar_i / abar_{prev kept}; also the kept-index map."""
    ac = np.cumprod(1.0 - np.asarray(base_betas, dtype=np.float64))
    keep = set(use_timesteps)
    last, nb, tmap = 1.0, [], []
    for i, v in enumerate(ac):
        if i in keep:
            nb.append(1 - v / last)
            last = v
            tmap.append(i)
    return np.array(nb), tmap


def make_schedule(steps=1000, noise_schedule="linear", timestep_respacing=""):
    """script_util.py:284-326 (schedule part): -> (tables dict, timestep_map, original T)."""
    base = named_betas(noise_schedule, steps)
    spec = timestep_respacing if timestep_respacing else [steps]
    nb, tmap = respaced_betas(base, space_timesteps(steps, spec))
    return tables(nb), tmap, steps
