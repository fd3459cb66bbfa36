import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from musediffusion_amd import synthetic, training
from musediffusion_amd.models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps
from musediffusion_amd.models.network import TransformerNetModel
from test_dropout_gpu import mask_to_bits
from test_training_gpu import CpuDraws
DEV = "cuda"
torch.manual_seed(3)
E, H, B, V, L, p = 32, 128, 2, 97, 512, 0.1
m = TransformerNetModel(E, E, 32, V, L, dropout=p, bert_hidden=H, bert_layers=2, bert_heads=2, bert_ffn=256, compute_dtype="bf16",
                        bert_hidden_dropout=p, bert_attention_dropout=p)
m.train().requires_grad_(True).to(DEV)
diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                       rescale_timesteps=True, predict_xstart=True)
gen = torch.Generator().manual_seed(9)
full = {"emb": (torch.rand(B * L, H, generator=gen) >= p).to(torch.uint8).to(DEV)}
for i in range(2):
    full["l%d.attn" % i] = mask_to_bits((torch.rand(B * 2, L, L, generator=gen) >= p).to(DEV))
    full["l%d.ao" % i] = (torch.rand(B * L, H, generator=gen) >= p).to(torch.uint8).to(DEV)
    full["l%d.ffn" % i] = (torch.rand(B * L, H, generator=gen) >= p).to(torch.uint8).to(DEV)
batch = {k: v % V for k, v in synthetic.training_batch(B, L, seed=4).items() if k != "length"}
batch["input_mask"] = synthetic.training_batch(B, L, seed=4)["input_mask"]
t = torch.tensor([100, 1500], device=DEV)
names = [n for n, _ in m.named_parameters()]

def run(fa, ff, sites):
    training.FUSED_ATTENTION, training.FUSED_FFN = fa, ff
    m.dropout_masks = {k: v for k, v in full.items() if any(k.endswith(s) or k == s for s in sites)}
    m.bert_hidden_dropout = p if any(s in ("ao", "ffn") for s in sites) else 0.0
    m.bert_attention_dropout = p if "attn" in sites else 0.0
    m.dropout.p = p if "emb" in sites else 0.0
    # sites not listed but with nonzero rate would use philox; keep rates zero for them by class
    m.zero_grad(set_to_none=True)
    with CpuDraws(5):
        terms = diff.training_losses(m, t, model_kwargs=batch)
    terms["loss"].mean().backward()
    return terms["loss"].detach().cpu(), [q.grad.detach().float().cpu().clone() for q in m.parameters()]


def top(g):
    return sorted(((float(x.abs().max()), n) for x, n in zip(g, names)), reverse=True)[:4]

for sites in ([], ["emb"]):
    la, ga = run(True, False, sites)
    lb, gb = run(True, False, sites)
    lc, gc = run(False, False, sites)
    ld, gd = run(False, False, sites)
    cat = lambda g: torch.cat([x.flatten() for x in g])
    cs = lambda a, b: float(torch.nn.functional.cosine_similarity(cat(a), cat(b), dim=0))
    print(sites, "fused/fused %.6f  unf/unf %.6f  fused/unf %.6f" % (cs(ga, gb), cs(gc, gd), cs(ga, gc)))
    print("  fused top", top(ga))
    print("  unfus top", top(gc))
    for x, y, n in zip(ga, gc, names):
        c = float(torch.nn.functional.cosine_similarity(x.flatten(), y.flatten(), dim=0))
        if c < 0.98:
            print("   ", n, "cos %.4f" % c, "absmax %.3e %.3e" % (float(x.abs().max()), float(y.abs().max())))
