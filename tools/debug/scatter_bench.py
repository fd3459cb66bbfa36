import torch, sys, os
sys.path.insert(0, os.getcwd())
from musediffusion_amd import _lib, synthetic
from musediffusion_amd._lib import lib, check, current_stream
dev="cuda"
b = synthetic.training_batch(32, 1024, seed=1)
ids = b["input_ids"].reshape(-1).to(torch.int32).to(dev)
V,E = synthetic.VOCAB_SIZE, 128
g = torch.randn(ids.numel(), E, device=dev)
dW = torch.zeros(V, E, device=dev)
ws = torch.empty(int(lib().mh_scatter_add_rows_workspace_bytes(E, V)), device=dev, dtype=torch.uint8)
def run():
    check(lib().mh_scatter_add_rows(g.data_ptr(), ids.data_ptr(), dW.data_ptr(), ids.numel(), E, V, ws.data_ptr(), ws.numel(), current_stream()))
run(); torch.cuda.synchronize()
ref = torch.zeros(V, E, device=dev).index_add_(0, ids.long(), g)
print("max err", float((dW - ref).abs().max()), "top id share", float(torch.bincount(ids.long()).max())/ids.numel())
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("us per call", e0.elapsed_time(e1)/50*1e3)
