#!/usr/bin/env python3
"""Phase profile of the attention backward dQ kernel (debug build of attention_bwd.hip with -DMH_BWD_PROF, loaded through MUSEHIP_AB=1
MUSEHIP_LIB=...): per wave, the shader clocks spent waiting for a stage's DMA, at the barrier behind it, in the stage's work and at the
end-of-stage barrier.  The dropout-free kernel is profiled (its keep_bits argument carries the buffer)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from musediffusion_amd._lib import check, current_stream, lib, ptr  # noqa: E402

B, L, nh, dh = 32, 1024, 8, 64
H = nh * dh
dev = "cuda"
qkv = (torch.randn(B * L, 3 * H, device=dev) * 0.7).to(torch.bfloat16)
vt = torch.zeros(B * nh * dh * L + 256, device=dev, dtype=torch.bfloat16)
L_, st = lib(), current_stream()
check(L_.mh_head_permute(qkv.data_ptr() + 2 * H * 2, ptr(vt), 3 * H, B, L, nh, dh, 3, 1, st))
out = torch.empty(B * L, H, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B * nh * L, device=dev, dtype=torch.float32)
check(L_.mh_attention_stream_fwd_drop(qkv.data_ptr(), qkv.data_ptr() + H * 2, ptr(vt), ptr(out), H, 0, B, L, nh, dh, 1 / math.sqrt(dh), ptr(lse), L * 3 * H, dh,
                                      3 * H, None, None, 0, st))
dctx = (torch.randn(B * L, H, device=dev) * 0.1).to(torch.bfloat16)
dqkv = torch.empty(B * L, 3 * H, device=dev, dtype=torch.bfloat16)
Dv = torch.empty(B * nh * L, device=dev, dtype=torch.float32)
buf = torch.zeros(256 * 8 * 4, device=dev, dtype=torch.int64)
for rep in range(3):
    buf.zero_()
    check(L_.mh_attention_stream_bwd_drop(qkv.data_ptr(), qkv.data_ptr() + H * 2, qkv.data_ptr() + 2 * H * 2, None, None, ptr(dctx), None, ptr(out), ptr(lse),
                                          ptr(Dv), dqkv.data_ptr(), dqkv.data_ptr() + H * 2, dqkv.data_ptr() + 2 * H * 2, 3 * H, B, L, nh, dh, 1 / math.sqrt(dh),
                                          L * 3 * H, dh, 3 * H, L * H, dh, H, buf.data_ptr(), 0.0, st))
    torch.cuda.synchronize()
    t = buf.view(256, 8, 4).double().cpu()
    print("dQ kernel, per wave (shader clocks, mean over 256 blocks): DMA wait %.0f  top barrier %.0f  stage work %.0f  end barrier %.0f  (sum %.0f)"
          % (*[float(t[:, :, i].mean()) for i in range(4)], float(t.sum(-1).mean())))
    print("  by wave index: work     " + " ".join("%6.0f" % float(t[:, w, 2].mean()) for w in range(8)))
    print("  by wave index: barriers " + " ".join("%6.0f" % float((t[:, w, 1] + t[:, w, 3]).mean()) for w in range(8)))
    print("  by wave index: DMA wait " + " ".join("%6.0f" % float(t[:, w, 0].mean()) for w in range(8)))
