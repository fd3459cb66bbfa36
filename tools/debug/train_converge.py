#!/usr/bin/env python3
"""Sanity run of the training path as a whole: N optimizer steps on ONE fixed synthetic batch (d_model 512, 4 layers, seq_len 512,
train mode, dropout 0.1, bf16) - the loss must fall - with the round's tape switches on and off (same seed, same batch): the two
trajectories must stay close (they differ by bf16 rounding of the stored GELU derivative and by nothing else)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from musediffusion_amd import synthetic, training  # noqa: E402
from musediffusion_amd.models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps  # noqa: E402
from musediffusion_amd.models.network import TransformerNetModel  # noqa: E402
from musediffusion_amd.models.step_sample import FixSampler  # noqa: E402
from musediffusion_amd.train_step import TrainStep  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda"
batch = {k: v.to(dev) for k, v in synthetic.training_batch(8, 512, seed=3).items()}


def run(switches):
    for k, v in switches.items():
        setattr(training, k, v)
    torch.manual_seed(0)
    np.random.seed(0)
    m = TransformerNetModel(128, 128, 128, synthetic.VOCAB_SIZE, 512, dropout=0.1, bert_hidden=512, bert_layers=4, bert_heads=8, bert_ffn=2048,
                            compute_dtype="bf16").to(dev).train()
    d = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000), rescale_timesteps=True,
                        predict_xstart=True)
    ts = TrainStep(m, d, microbatch=-1, lr=3e-4, ema_rate="0.99", schedule_sampler=FixSampler(d))
    out = []
    for i in range(N):
        losses, gn = ts.run_step(batch)
        out.append(float(losses["loss"].mean()))
    return out


a = run(dict(GELU_DERIV_FWD=True, LN_BWD_DROP=True))
b = run(dict(GELU_DERIV_FWD=False, LN_BWD_DROP=False))
training.GELU_DERIV_FWD, training.LN_BWD_DROP = True, True
for i in range(0, N, max(1, N // 10)):
    print("step %3d: loss %.4f (round-3 tape)  %.4f (round-2 forms)" % (i, a[i], b[i]))
first, last_a, last_b = sum(a[:5]) / 5, sum(a[-10:]) / 10, sum(b[-10:]) / 10
print("mean of the first 5 steps %.4f; of the last 10: %.4f (round-3 tape) %.4f (round-2 forms); largest per-step difference %.2e"
      % (first, last_a, last_b, max(abs(x - y) for x, y in zip(a, b))))
assert last_a < 0.7 * first, "the loss does not fall"
assert max(abs(x - y) for x, y in zip(a, b)) < 0.02, "the two tapes diverge"
