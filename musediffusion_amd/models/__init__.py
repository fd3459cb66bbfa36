"""Mirror of the reference's `MuseDiffusion.models` package (network / diffusion / rounding /
step_sample) plus the alias modules BASELINE.json's north_star names (denoising_model,
gaussian_diffusion, nn)."""
