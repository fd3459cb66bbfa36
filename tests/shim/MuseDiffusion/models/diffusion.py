# MuseDiffusion/models/diffusion.py
from musediffusion_amd.models.diffusion import (GaussianDiffusion, SpacedDiffusion, _WrappedModel,
    _extract_into_tensor, betas_for_alpha_bar, get_named_beta_schedule, mean_flat, space_timesteps, unwrap_model)
