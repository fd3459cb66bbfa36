# MuseDiffusion/models/rounding.py
from musediffusion_amd.models.rounding import denoised_fn_round, get_efficient_knn, get_knn
