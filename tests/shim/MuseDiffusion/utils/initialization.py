"""Stand-in for the reference's factory module on machines that do not hold the reference checkout (the GPU box).  The recipe of
INTEGRATION.md does NOT replace MuseDiffusion/utils/initialization.py: the reference's own file keeps working because its only
contact with the hot path is two imports inside create_model_and_diffusion (utils/initialization.py:110-112), which this
stand-in issues through the same `MuseDiffusion.models.*` names.  tests/test_shim_cpu.py runs the reference's REAL file over this shim
where /root/reference exists.  The optional bert_* / compute_dtype fields only let a test reach the small golden shapes."""


def create_model_and_diffusion(args):
    from MuseDiffusion.models.diffusion import SpacedDiffusion, space_timesteps, get_named_beta_schedule
    from MuseDiffusion.models.network import TransformerNetModel
    extra = {k: getattr(args, k) for k in ("bert_hidden", "bert_layers", "bert_heads", "bert_ffn", "compute_dtype") if hasattr(args, k)}
    model = TransformerNetModel(input_dims=args.hidden_dim, output_dims=args.hidden_dim, hidden_t_dim=args.hidden_t_dim,
                                vocab_size=args.vocab_size, seq_len=args.seq_len, dropout=args.dropout, **extra)
    respacing = args.timestep_respacing or [args.diffusion_steps]
    diffusion = SpacedDiffusion(use_timesteps=space_timesteps(args.diffusion_steps, respacing),
                                betas=get_named_beta_schedule(args.noise_schedule, args.diffusion_steps),
                                rescale_timesteps=args.rescale_timesteps, predict_xstart=args.predict_xstart)
    return model, diffusion
