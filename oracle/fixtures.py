"""Oracle-side helpers shared by tools/make_golden.py and the tests: the seeded *inputs* of
every golden case.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Inputs are derived from isolated `torch.Generator` seeds so that a test can rebuild them
without the fixture having to store them (the config-1 fixture keeps only strided slices of
the outputs to stay small); the tiny fixtures store them too, as a drift detector.
"""
import torch

from . import denoiser

CONFIGS = {
    "tiny": dict(H=64, nL=2, nh=4, F=256, E=32, Tt=32, L=16, B=2, V=729),
    "same": dict(H=64, nL=1, nh=2, F=128, E=64, Tt=16, L=24, B=3, V=97),      # E == H: no up/down proj
    "c1": dict(H=128, nL=2, nh=4, F=512, E=128, Tt=128, L=128, B=8, V=729),   # BASELINE config 1 shape
}
CONFIGS.update({
    # the only shape the reference itself can instantiate (bert-base: H 768, 12 heads of 64, ffn 3072; network.py:44-46), two layers
    "bb": dict(H=768, nL=2, nh=12, F=3072, E=128, Tt=128, L=64, B=2, V=729),
    # ... with the released checkpoints' embedding width (README.md:534: "embedding dim 500")
    "bb500": dict(H=768, nL=2, nh=12, F=3072, E=500, Tt=128, L=32, B=2, V=729),
})
CONFIGS.update({
    # the BENCHMARKED shapes at two layers / two sequences (round 3): BASELINE config 2's width and seq_len 512 - the streaming
    # attention, the 128x512 full-row tile and the 256x128 tiles of the bf16 path run under outputs of the reference itself ...
    "c2s": dict(H=512, nL=2, nh=8, F=2048, E=128, Tt=128, L=512, B=2, V=729),
    # ... and config 5's seq_len 1024 (training_losses both variants, dropout 0 and injected masks)
    "c5s": dict(H=512, nL=2, nh=8, F=2048, E=128, Tt=128, L=1024, B=2, V=729),
})
CONFIGS.update({
    # round 5: the TRUE DEPTH of the benchmarked / reference-true encoders (12 layers), so that the error growth of every compute mode
    # through twelve post-LN layers is held against outputs of the reference itself, not against another mode of this library
    "c2d": dict(H=512, nL=12, nh=8, F=2048, E=128, Tt=128, L=512, B=2, V=729),     # BASELINE config 2's denoiser, 2 sequences
    "bbd": dict(H=768, nL=12, nh=12, F=3072, E=128, Tt=128, L=64, B=2, V=729),     # bert-base as network.py:44-46 builds it
    "c5d": dict(H=512, nL=12, nh=8, F=2048, E=128, Tt=128, L=1024, B=2, V=729),    # BASELINE config 5's denoiser (training_losses), 2 sequences
})
SEEDS = {"tiny": 0, "same": 1, "c1": 2, "bb": 3, "bb500": 4, "c2s": 5, "c5s": 6, "c2d": 7, "bbd": 8, "c5d": 9}
COMPACT = ("c1", "bb", "bb500", "c2s", "c2d", "bbd")   # model fixtures that keep every 8th position of [B, L, *] outputs and no inputs
HIDDEN_KEEP = {"c2d": (0, 3, 7, 11), "bbd": (0, 3, 7, 11)}   # deep fixtures record these layers' outputs only


def slim(t):
    """What a 'slim' loss fixture keeps of a gradient: every 4th row and column of a matrix with more than 64 K elements."""
    return t[::4, ::4] if (t.dim() == 2 and t.numel() > 65536) else t
DROPOUT_P = 0.1   # the reference's train-mode rate at all three kinds of site (config/train.py:61, bert-base config)
EMB_STD = 0.5
NOISING_T = 150


def seeded_randn(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(int(seed)))


def state_dict(tag):
    c = CONFIGS[tag]
    return denoiser.random_state_dict(c["E"], c["H"], c["F"], c["nL"], c["V"], c["L"], c["Tt"],
                                      seed=SEEDS[tag], emb_std=EMB_STD)


def token_batch(tag):
    from musediffusion_amd import synthetic  # data-only generator (no kernels)
    c = CONFIGS[tag]
    batch = synthetic.training_batch(c["B"], c["L"], seed=SEEDS[tag] + 1)
    if c["V"] < synthetic.VOCAB_SIZE:
        for k in ("input_ids", "correct_ids"):
            batch[k] = batch[k] % c["V"]
    return batch


def case_inputs(tag, emb_weight):
    """All deterministic inputs of a model case.  emb_weight = word_embedding.weight [V,E]."""
    c, s = CONFIGS[tag], SEEDS[tag]
    B, L, E = c["B"], c["L"], c["E"]
    batch = token_batch(tag)
    x_start = emb_weight[batch["correct_ids"]]
    inp = dict(batch=batch, x_start=x_start)
    inp["mask3"] = torch.broadcast_to(batch["input_mask"].unsqueeze(-1), x_start.shape)
    inp["fwd_x"] = seeded_randn(1000 + s, B, L, E)
    inp["fwd_t"] = torch.tensor([(7.0 + 131 * b) % 1000 for b in range(B)]) * 0.5
    inp["round_in"] = x_start + 0.3 * seeded_randn(1001 + s, B, L, E)
    inp["gen_noise0"] = seeded_randn(1002 + s, B, L, E)
    inp["mod_noise"] = seeded_randn(1003 + s, B, L, E)
    inp["q_t"] = torch.tensor([(3 + 577 * b) % 2000 for b in range(B)])
    inp["q_noise"] = seeded_randn(1004 + s, B, L, E)
    inp["free_t"] = torch.tensor([(11 + 397 * b) % 2000 for b in range(B)])
    return inp


def loss_inputs(tag):
    c, s = CONFIGS[tag], SEEDS[tag]
    B = c["B"]
    return dict(batch=token_batch(tag),
                t=torch.tensor([0] + [(17 + 613 * b) % 2000 for b in range(1, B)]),
                w=torch.linspace(0.5, 1.5, B))


# seeds handed to torch.manual_seed() right before a reference call that draws from the
# global generator (p_sample / ddim_sample / loops / training_losses)
def step_seed(tag):
    return 600 + SEEDS[tag]


def free_seed(tag):
    return 601 + SEEDS[tag]


def loop_seed(tag, which):
    return {"ddim50": 700, "p12": 701, "mod": 702}[which] + SEEDS[tag]


def loss_seed(tag):
    return 800 + SEEDS[tag]


def dropout_sites(tag):
    """Site names in the order the reference's forward reaches its dropouts (network.py:149, then per BertLayer: attention
    probabilities, attention-output dense, FFN-output dense) with the shape of the tensor each one masks."""
    c = CONFIGS[tag]
    B, L, H, nh = c["B"], c["L"], c["H"], c["nh"]
    sites = [("emb", (B, L, H))]
    for i in range(c["nL"]):
        sites += [("l%d.attn" % i, (B, nh, L, L)), ("l%d.ao" % i, (B, L, H)), ("l%d.ffn" % i, (B, L, H))]
    return sites


def dropout_masks(tag, p=DROPOUT_P, seed_base=900):
    """Deterministic keep masks (bool) for every dropout site of one forward: Bernoulli(1 - p) from an isolated generator."""
    g = torch.Generator().manual_seed(seed_base + SEEDS[tag])
    return {name: torch.rand(shape, generator=g) >= p for name, shape in dropout_sites(tag)}
