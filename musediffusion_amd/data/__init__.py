"""Device-side batch producers: the reference's data/corruption.py and data/wrapper.py collate,
running on ragged int32 token tensors that already live on the GPU."""
from .corruption import Corruptions, masking_note, masking_token, random_rotating, randomize_note  # noqa: F401
from .wrapper import collate_batches, to_ragged  # noqa: F401
