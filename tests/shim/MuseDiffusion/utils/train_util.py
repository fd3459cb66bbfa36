"""MuseDiffusion/utils/train_util.py as a re-export: run/train.py:26 imports TrainLoop from here."""
from musediffusion_amd.utils.train_util import TrainLoop, update_ema  # noqa: F401
