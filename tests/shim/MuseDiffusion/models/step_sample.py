# MuseDiffusion/models/step_sample.py
from musediffusion_amd.models.step_sample import (create_named_schedule_sampler, LossAwareSampler,
    LossSecondMomentResampler, ScheduleSampler, UniformSampler, FixSampler)
