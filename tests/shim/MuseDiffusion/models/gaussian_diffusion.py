# north_star alias (BASELINE.json): MuseDiffusion.models.gaussian_diffusion.GaussianDiffusion
from musediffusion_amd.models.gaussian_diffusion import (GaussianDiffusion, SpacedDiffusion, _WrappedModel, _extract_into_tensor,
    betas_for_alpha_bar, get_named_beta_schedule, mean_flat, space_timesteps, unwrap_model)
