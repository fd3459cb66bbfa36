"""Alias module: north_star name `models.denoising_model.TransformerNetModel`."""
from .network import TransformerNetModel  # noqa: F401
