"""Alias module: north_star name `models.gaussian_diffusion.GaussianDiffusion`."""
from .diffusion import (GaussianDiffusion, SpacedDiffusion, _WrappedModel, _extract_into_tensor,  # noqa: F401
                        betas_for_alpha_bar, get_named_beta_schedule, mean_flat, space_timesteps, unwrap_model)
