# MuseDiffusion/models/network.py
from musediffusion_amd.models.network import TransformerNetModel            # network.py:20
