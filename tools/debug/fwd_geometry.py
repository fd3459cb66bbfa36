import sys, os, math, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from musediffusion_amd import _lib
_lib.use_debug_library()   # the A/B switches live in libmusehip_dbg.so (include/musehip_dbg.h)
from musediffusion_amd._lib import check, current_stream, lib, ptr
B, L, nh, dh = 32, 1024, 8, 64
H = nh * dh
dev = "cuda"
qkv = (torch.randn(B * L, 3 * H, device=dev) * 0.7).to(torch.bfloat16)
vt = torch.zeros(B * nh * dh * L + 256, device=dev, dtype=torch.bfloat16)
L_ = lib(); st = current_stream()
check(L_.mh_head_permute(qkv.data_ptr() + 2 * H * 2, ptr(vt), 3 * H, B, L, nh, dh, 3, 1, st))
out = torch.empty(B * L, H, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B * nh * L, device=dev, dtype=torch.float32)
def fwd():
    check(L_.mh_attention_stream_fwd_drop(qkv.data_ptr(), qkv.data_ptr() + H * 2, ptr(vt), ptr(out), H, 0, B, L, nh, dh, 1 / math.sqrt(dh), ptr(lse), L * 3 * H, dh, 3 * H, None, None, 0, st))
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(3):
    for mode in (1, 2):
        L_.mh_attention_set_stream(mode)
        print("mode %d (%s): no dropout %.1f us" % (mode, "16 waves x 256-key stages" if mode == 1 else "8 waves x 128-key stages", t(fwd)))
L_.mh_attention_set_stream(1)
