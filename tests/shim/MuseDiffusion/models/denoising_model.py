# north_star alias (BASELINE.json): MuseDiffusion.models.denoising_model.TransformerNetModel
from musediffusion_amd.models.denoising_model import TransformerNetModel
