# north_star alias (BASELINE.json): MuseDiffusion.models.nn.timestep_embedding
from musediffusion_amd.models.nn import timestep_embedding
