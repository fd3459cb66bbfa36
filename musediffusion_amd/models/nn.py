"""Alias module: north_star name `models.nn.timestep_embedding` (the reference keeps it as a
staticmethod, MuseDiffusion/models/network.py:108-129)."""
from .network import TransformerNetModel

timestep_embedding = TransformerNetModel.timestep_embedding
