"""Factory of the hot path's two objects, API of MuseDiffusion/utils/initialization.py:108-136."""
import random

import numpy as np
import torch


def seed_all(seed, deterministic=False, rank=0):
    """utils/initialization.py:11-26 (rank offset unless deterministic)."""
    seed = hash(seed) if deterministic else hash(seed) + rank
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)


def create_model_and_diffusion(args, **model_overrides):
    """Build (TransformerNetModel, SpacedDiffusion) from a settings object with the reference's
    field names (hidden_dim, hidden_t_dim, vocab_size, seq_len, dropout, noise_schedule,
    diffusion_steps, timestep_respacing, rescale_timesteps, predict_xstart).

    Optional fields / overrides the reference does not have: bert_hidden, bert_layers, bert_heads,
    bert_ffn (Transformer shape; default bert-base-uncased like network.py:44) and compute_dtype."""
    from ..models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps
    from ..models.network import TransformerNetModel

    extra = {}
    for k in ("bert_hidden", "bert_layers", "bert_heads", "bert_ffn", "bert_hidden_dropout", "bert_attention_dropout", "compute_dtype"):
        if k in model_overrides:
            extra[k] = model_overrides[k]
        elif getattr(args, k, None) is not None:
            extra[k] = getattr(args, k)
    model = TransformerNetModel(input_dims=args.hidden_dim, output_dims=args.hidden_dim,
                                hidden_t_dim=args.hidden_t_dim, vocab_size=args.vocab_size, seq_len=args.seq_len,
                                dropout=args.dropout, **extra)
    betas = get_named_beta_schedule(args.noise_schedule, args.diffusion_steps)
    timestep_respacing = args.timestep_respacing or [args.diffusion_steps]
    diffusion = SpacedDiffusion(use_timesteps=space_timesteps(args.diffusion_steps, timestep_respacing), betas=betas,
                                rescale_timesteps=args.rescale_timesteps, predict_xstart=args.predict_xstart)
    return model, diffusion
