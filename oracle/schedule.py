"""Oracle: beta schedules, float64 diffusion tables and timestep respacing.

TEST INFRASTRUCTURE (see oracle/__init__.py).  numpy float64 throughout, as
the reference keeps its tables (MuseDiffusion/models/diffusion.py:146-183).
"""
import math
from types import SimpleNamespace

import numpy as np


def _discretize(T, alpha_bar, max_beta=0.999, shifted=False):
    """beta_i = min(1 - abar((i+1)/T)/abar(i/T), max_beta)  (diffusion.py:101-118).

    ``shifted`` is the 'left' variant (diffusion.py:80-98): first beta is
    min(1-abar(0), max_beta) and only T-1 ratio terms follow.
    """
    out = []
    if shifted:
        out.append(min(1 - alpha_bar(0), max_beta))
    n = T - 1 if shifted else T
    for i in range(n):
        out.append(min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), max_beta))
    return np.array(out)


def named_betas(name, T):
    """diffusion.py:22-77 (get_named_beta_schedule)."""
    scale = 1000 / T
    if name == "linear":
        return np.linspace(scale * 1e-4, scale * 0.02, T, dtype=np.float64)
    if name == "cosine":
        return _discretize(T, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    if name == "sqrt":
        return _discretize(T, lambda t: 1 - np.sqrt(t + 0.0001))
    if name == "trunc_cos":
        return _discretize(T, lambda t: np.cos((t + 0.1) / 1.1 * np.pi / 2) ** 2, shifted=True)
    if name == "trunc_lin":
        return np.linspace(scale * 1e-4 + 0.01, scale * 0.02 + 0.01, T, dtype=np.float64)
    if name == "pw_lin":
        head = np.linspace(scale * 1e-4 + 0.01, scale * 1e-4, 10, dtype=np.float64)
        tail = np.linspace(scale * 1e-4, scale * 0.02, T - 10, dtype=np.float64)
        return np.concatenate([head, tail])
    raise NotImplementedError("unknown beta schedule: {}".format(name))


TABLE_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
)


def tables(betas):
    """The 13 float64 tables of GaussianDiffusion.__init__ (diffusion.py:146-183)."""
    b = np.array(betas, dtype=np.float64)
    assert b.ndim == 1 and (b > 0).all() and (b <= 1).all()
    a = 1.0 - b
    ac = np.cumprod(a, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    ac_next = np.append(ac[1:], 0.0)
    pv = b * (1.0 - ac_prev) / (1.0 - ac)
    t = SimpleNamespace(
        betas=b,
        alphas_cumprod=ac,
        alphas_cumprod_prev=ac_prev,
        alphas_cumprod_next=ac_next,
        sqrt_alphas_cumprod=np.sqrt(ac),
        sqrt_one_minus_alphas_cumprod=np.sqrt(1.0 - ac),
        log_one_minus_alphas_cumprod=np.log(1.0 - ac),
        sqrt_recip_alphas_cumprod=np.sqrt(1.0 / ac),
        sqrt_recipm1_alphas_cumprod=np.sqrt(1.0 / ac - 1),
        posterior_variance=pv,
        posterior_log_variance_clipped=np.log(np.append(pv[1], pv[1:])),
        posterior_mean_coef1=b * np.sqrt(ac_prev) / (1.0 - ac),
        posterior_mean_coef2=(1.0 - ac_prev) * np.sqrt(a) / (1.0 - ac),
        num_timesteps=int(b.shape[0]),
    )
    # "fixed large" sampling variance, rebuilt per step by the reference (diffusion.py:313-314)
    t.model_variance = np.append(pv[1], b[1:])
    t.model_log_variance = np.log(t.model_variance)
    return t


def space_timesteps(T, section_counts):
    """diffusion.py:920-969."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, T):
                if len(range(0, T, stride)) == want:
                    return set(range(0, T, stride))
            raise ValueError("cannot create exactly {} steps with an integer stride".format(T))
        section_counts = [int(x) for x in section_counts.split(",")]
    per, extra = divmod(T, len(section_counts))
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError("cannot divide section of {0} steps into {1}".format(size, cnt))
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


def spaced(betas, use_timesteps):
    """SpacedDiffusion.__init__ (diffusion.py:981-996): recomputed betas + timestep_map."""
    use = set(use_timesteps)
    base = tables(betas)
    last, new_betas, tmap = 1.0, [], []
    for i, ac in enumerate(base.alphas_cumprod):
        if i in use:
            new_betas.append(1 - ac / last)
            last = ac
            tmap.append(i)
    t = tables(np.array(new_betas))
    t.timestep_map = tmap
    t.original_num_steps = len(betas)
    return t


def make_diffusion(noise_schedule="sqrt", diffusion_steps=2000, timestep_respacing="",
                   rescale_timesteps=True, predict_xstart=True):
    """utils/initialization.py:123-134 (the diffusion half of create_model_and_diffusion)."""
    betas = named_betas(noise_schedule, diffusion_steps)
    resp = timestep_respacing if timestep_respacing else [diffusion_steps]
    d = spaced(betas, space_timesteps(diffusion_steps, resp))
    d.rescale_timesteps = rescale_timesteps
    d.predict_xstart = predict_xstart
    return d
